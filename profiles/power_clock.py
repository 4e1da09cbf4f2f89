"""Board power and shader clock while the hot kernels run (run on the GPU box):  python profiles/power_clock.py [seconds_per_leg]

Is the dense scan (and the encoder GEMM) limited by its schedule or by the chip's power management?  The 2.5 PFLOP/s MFMA
peak and the guide's per-CU rates assume ~2.4 GHz; under sustained MFMA + HBM load the chip lowers the shader clock to stay
inside its power limit.  For every leg this samples, every 50 ms from a side thread while the leg loops on the GPU:
  * socket power (hwmon power1_average / power1_input, or `amd-smi metric -p`), the power cap (power1_cap),
  * shader clock as the driver reports it (hwmon freq1_input, or pp_dpm_sclk's starred level / amd-smi),
and takes from the library the clock the scan kernel measured ITSELF (cycles per 100 MHz tick over its tile loops,
bh_counters.shader_mhz).
Legs: idle; dense scan production / no filter / MFMA + rendezvous only / stream only (bench-only ablations of the 256-query
kernel: results invalid, timings meaningful); the BERT-base forward pass of bench.py's encoder leg.
Prints one JSON object."""
import json
import os
import statistics
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import bergen_amd  # noqa: E402
from bergen_amd import _lib  # noqa: E402


Sampler = bench.PowerSampler


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    _lib.init(0)
    dev = torch.device("cuda", 0)
    smp = Sampler()
    res = {"sampler": smp.source, "pci": smp.pci, "power_cap_watts": smp.cap_watts(), "seconds_per_leg": secs, "legs": {}}

    smp.start()
    time.sleep(2.0)
    res["legs"]["idle"] = smp.stop(0.0)

    dim, k, n_total = 768, 50, 21_000_000
    q = bench.make_queries(2837, dim, dev)
    ix = bergen_amd.FlatIndex(n_total, dim, metric="ip", device=0)
    bench.fill_shard(ix, 0, n_total, dim, q, n_total, dev)
    ix.finalize()
    q256 = q[:256].contiguous()
    for name, abl in (("scan_production", 0), ("scan_no_filter", 1), ("scan_no_filter_no_lds_reads", 3), ("scan_no_filter_no_refill", 9),
                      ("scan_mfma_and_rendezvous_only", 11), ("scan_stream_only", 7), ("scan_production_again", 0)):
        _lib.set_option("ablate", abl)
        _lib.set_option("certify", 0 if abl else 1)  # (an ablated scan proves nothing: no fall-back passes behind it)
        ix.search(q256, k)
        ms, mhz = [], []
        smp.start()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < secs:
            for _ in range(20):
                ix.search(q256, k)
                c = ix.counters()
                ms.append(c["scan_ms"] / c["n_passes"])
                mhz.append(c["shader_mhz"])
        leg = smp.stop()
        leg.update(scan_ms_per_pass_median=round(statistics.median(ms), 4), scan_ms_per_pass_last_quarter=round(statistics.median(ms[-len(ms) // 4:]), 4),
                   kernel_measured_shader_mhz=round(statistics.median(mhz)), launches=len(ms))
        res["legs"][name] = leg
    _lib.set_option("ablate", 0)
    _lib.set_option("certify", 1)
    for name, sk, nq in (("scan192_production", 2, 192), ("scan128_production", 0, 128)):
        _lib.set_option("scan_kernel", sk)
        qn = q[:nq].contiguous()
        ix.search(qn, k)
        ms = []
        smp.start()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < secs:
            for _ in range(20):
                ix.search(qn, k)
                c = ix.counters()
                ms.append(c["scan_ms"] / c["n_passes"])
        leg = smp.stop()
        leg.update(scan_ms_per_pass_median=round(statistics.median(ms), 4), queries_per_pass=nq, launches=len(ms))
        res["legs"][name] = leg
    _lib.set_option("scan_kernel", 3)
    ix.close()
    del ix
    torch.cuda.empty_cache()

    # the encoder forward pass of bench.py's leg (BERT-base, 512 passages per step)
    from bergen_amd import BertEncoder, synth
    cfg = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
               max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")
    sd = synth.random_bert(cfg, seed=31)
    enc = BertEncoder(cfg, {k_: torch.from_numpy(v) for k_, v in sd.items()}, device=0)
    rng = np.random.default_rng(6)
    lens = np.clip(np.rint(rng.normal(130, 30, size=512)), 16, 256).astype(np.int64)
    T = int(lens.max())
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.int64)
    ids = rng.integers(1, cfg["vocab_size"], size=(512, T)).astype(np.int64) * mask
    kw = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}
    enc.encode_pooled(kw, "cls")
    torch.cuda.synchronize()
    n = 0
    smp.start()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < secs:
        for _ in range(10):
            enc.encode_pooled(kw, "cls")
        torch.cuda.synchronize()
        n += 10
    dt = time.perf_counter() - t0
    leg = smp.stop()
    leg.update(forward_ms=round(dt / n * 1e3, 3), passages_per_s=round(512 * n / dt))
    res["legs"]["encoder_forward"] = leg
    print(json.dumps(res))


if __name__ == "__main__":
    main()
