#!/usr/bin/env python
"""Kernel-level measurements of the bi-encoder forward pass on one MI355X (run on the GPU box):

  python profiles/bench_encoder.py [--out gpurun_out/encoder_bench.json] [--quick]

  * GEMM sweep: every tile configuration (gemm_f16.hip variants 1..7) on the five projection shapes of a
    bert-base layer at M = 66 560 packed tokens (512 sequences x ~130 tokens), TFLOP/s against the
    2.5 PFLOP/s dense fp16 MFMA peak;
  * attention kernel on the same batch;
  * whole forward pass (12 layers) per GEMM variant: passages/s, achieved TFLOP/s (algorithmic flops over
    real tokens), fraction of peak.
Synthetic token ids / lengths (SURVEY §8d "Encoder": lengths ~ clipped-Normal(130, 30) in [16, 256]) and
seeded random weights: no checkpoint exists offline.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0


def synth_batch(batch, vocab, seed=6, mean=130, std=30, lo=16, hi=256):
    rng = np.random.default_rng(seed)
    lens = np.clip(np.rint(rng.normal(mean, std, size=batch)), lo, hi).astype(np.int64)
    T = int(lens.max())
    ids = rng.integers(1, vocab, size=(batch, T)).astype(np.int64)
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.int64)
    return torch.from_numpy(ids * mask), torch.from_numpy(mask), lens


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "encoder_bench.json"))
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--no-gemm-sweep", action="store_true")
    args = ap.parse_args()
    from bergen_amd import BertEncoder, encoder
    from bergen_amd import synth

    dev = "cuda:0"
    res = {"permlane_mode": encoder.permlane_mode(), "gemm": [], "forward": []}
    g = torch.Generator(device=dev).manual_seed(0)
    M = 66560
    # (the residual adds of the attention-output / FFN-down projections live in the LayerNorm kernel)
    shapes = [("qk", M, 1536, 768, False, False), ("vt", 768, M, 768, False, False), ("attn_out", M, 768, 768, False, False),
              ("ffn1", M, 3072, 768, True, False), ("ffn2", M, 768, 3072, False, False)]
    variants = [5, 7, 8, 9]
    for name, m, n, k, gelu, resid in ([] if args.no_gemm_sweep else shapes):
        a = (torch.randn(m, k, generator=g, device=dev) * 0.5).half()
        w = (torch.randn(n, k, generator=g, device=dev) * 0.05).half()
        bias_mode = 2 if name == "vt" else 1
        bias = (torch.randn(m if bias_mode == 2 else n, generator=g, device=dev) * 0.1).half()
        r = (torch.randn(m, n, generator=g, device=dev) * 0.5).half() if resid else None
        out = torch.empty(m, n, dtype=torch.float16, device=dev)
        for v in variants:
            try:
                _, ms = encoder.gemm_f16(a, w, bias=bias, bias_mode=bias_mode, residual=r, gelu=gelu, variant=v, out=out,
                                         repeats=4 if args.quick else 11)
                tf = 2.0 * m * n * k / (ms * 1e-3) / 1e12
                res["gemm"].append({"shape": name, "M": m, "N": n, "K": k, "variant": v, "ms": ms, "tflops": tf,
                                    "frac_of_peak": tf / MFMA_PEAK_TFLOPS})
                print(f"gemm {name:8s} v{v}: {ms:8.3f} ms  {tf:7.1f} TF/s", flush=True)
            except Exception as e:  # keep sweeping
                res["gemm"].append({"shape": name, "variant": v, "error": str(e)})
                print(f"gemm {name} v{v}: ERROR {e}", flush=True)
        del a, w, out, r

    cfg = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
               max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")
    sd = synth.random_bert(cfg, seed=31)
    enc = BertEncoder(cfg, {k: torch.from_numpy(v) for k, v in sd.items()}, device=0)
    ids, mask, lens = synth_batch(args.batch, cfg["vocab_size"])
    kw = {"input_ids": ids, "attention_mask": mask}
    for v in [0] + ([] if args.quick else variants):
        enc.set_option("gemm_variant", v)
        enc.encode_pooled(kw, "cls")
        best, wall = 1e9, 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            enc.encode_pooled(kw, "cls")
            wall = min(wall, time.perf_counter() - t0)
            c = enc.counters()
            best = min(best, c["forward_ms"])
        tf = c["flops"] / (best * 1e-3) / 1e12
        row = {"gemm_variant": v, "batch": args.batch, "real_tokens": c["real_tokens"], "packed_rows": c["packed_rows"],
               "padded_tokens_reference": int(ids.numel()), "forward_ms": best, "wall_ms": wall * 1e3,
               "passages_per_s": args.batch / (best * 1e-3), "passages_per_s_wall": args.batch / wall,
               "tflops": tf, "frac_of_peak": tf / MFMA_PEAK_TFLOPS}
        res["forward"].append(row)
        print("forward", json.dumps(row), flush=True)
    enc.set_option("gemm_variant", 0)

    # attention alone on the same batch geometry
    off, cur = [], 0
    for n in lens:
        off.append(cur)
        cur = (cur + int(n) + 7) // 8 * 8
    rows = (cur + 32 + 255) // 256 * 256
    qk = (torch.randn(rows, 1536, generator=g, device=dev)).half()
    vt = (torch.randn(768, rows, generator=g, device=dev)).half()
    encoder.attention(qk, vt, off, lens.tolist(), 12, int(lens.max()))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        encoder.attention(qk, vt, off, lens.tolist(), 12, int(lens.max()))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    fl = 4.0 * float((lens.astype(np.float64) ** 2).sum()) * 768
    res["attention"] = {"ms_wall_incl_host": ms, "tflops": fl / (ms * 1e-3) / 1e12, "batch": args.batch}
    print("attention", res["attention"], flush=True)
    enc.close()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
