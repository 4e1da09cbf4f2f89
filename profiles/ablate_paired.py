"""In-kernel ablation ladder of the PAIRED headline instantiation (bh_scan_topk256_kernel<24, 64, 12, 3, 4, false, 128 | x, ...>: two
256-query passes per launch on partner workgroups, round-5 review item 4): the same 2 560-query search (five paired launches, no
remainder) over all 21 M x 768 rows with parts of the kernel compiled out (bench-only option `ablate`; results are INVALID for every
row but "production" and "unpaced"), interleaved rounds, per-launch time from the library's HIP events.
    python profiles/ablate_paired.py > profiles/r06_scan_paired_ablation.json
Rows: stream only (LDS-DMA ring + rendezvous + pacing) / + fragment reads / + MFMA (no filter) / production; MFMA + reads without the
refill; production without pacing (pair256 = 2) and unpaired (pair256 = 0: two launches of one pass each)."""
import json, os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, bergen_amd
from bergen_amd import _lib
_lib.init(0)
dev = torch.device("cuda", 0)
dim, k, nq, n = 768, 50, 2560, int(os.environ.get("ABL_ROWS", 21_000_000))
q = bench.make_queries(nq, dim, dev)
ix = bergen_amd.FlatIndex(n, dim, metric="ip", device=0)
bench.fill_shard(ix, 0, n, dim, q, n, dev)
ix.finalize()
rows = [("production", dict(ablate=0, pair256=1)), ("no_filter", dict(ablate=1, pair256=1)), ("stream_and_fragment_reads", dict(ablate=5, pair256=1)),
        ("stream_only", dict(ablate=7, pair256=1)), ("mfma_and_fragment_reads_no_refill", dict(ablate=9, pair256=1)),
        ("production_unpaced", dict(ablate=0, pair256=2)), ("production_unpaired", dict(ablate=0, pair256=0))]
t = {name: [] for name, _ in rows}
mhz = {name: [] for name, _ in rows}
for rnd in range(4):
    for name, opts in rows:
        for o, v in opts.items():
            ix.set_option(o, v)
        ix.set_option("certify", 0 if opts["ablate"] else 1)
        ix.search(q, k)
        c = ix.counters()
        if rnd:
            if c["paired_launches"]:
                t[name].append(c["paired_scan_ms"] / c["paired_launches"])
            else:  # unpaired: per PAIR of passes, for comparison
                t[name].append(2.0 * c["scan_ms"] / c["n_passes"])
            mhz[name].append(c["shader_mhz"])
flops = 2.0 * 512 * n * dim
out = {"workload": f"{nq} queries x {n} x {dim} fp16, top-{k}: five paired launches (2 x 256 queries each), lists of 64", "rows": []}
for name, opts in rows:
    ms = statistics.median(t[name])
    out["rows"].append({"variant": name, "options": opts, "ms_per_paired_launch": round(ms, 3), "ms_min": round(min(t[name]), 3),
                        "shader_mhz": round(statistics.median(mhz[name]), 0),
                        "mfma_tflops_if_all_flops_were_done": round(flops / (ms * 1e-3) / 1e12, 1), "valid_results": opts["ablate"] == 0})
    print(out["rows"][-1], file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
