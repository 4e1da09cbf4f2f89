#!/usr/bin/env python
"""Kernel ablations of the d=768 scan (bench-only).  Builds the bench corpus once, then times the
production kernel and the ABL variants (see scan_topk.hip).  Results -> gpurun_out/ablate.json"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa
import bergen_amd  # noqa
from bergen_amd import _lib  # noqa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 21_000_000
dev = torch.device("cuda:0")
_lib.init(0)
q = bench.make_queries(512, 768, dev)
ix = bergen_amd.FlatIndex(n, 768)
bench.fill_shard(ix, 0, n, 768, q, n, dev)
ix.finalize()
names = {0: "production", 1: "no epilogue", 2: "stream only (LDS-DMA + barriers)", 3: "LDS reads, no MFMA", 4: "MFMA, no LDS reads"}
out = []
for tile in (128,):
    for abl in (0, 1, 2, 3, 4, 0):
        _lib.set_option("query_tile", tile)
        _lib.set_option("ablate", abl)
        ix.search(q, 50)
        best = 1e9
        for _ in range(3):
            ix.search(q, 50)
            c = ix.counters()
            best = min(best, c["scan_ms"] / c["n_passes"])
        per = c["algorithmic_bytes"] / c["n_passes"]
        out.append({"tile": tile, "ablate": abl, "what": names[abl], "scan_ms_per_pass": best, "GBps": per / best / 1e6})
        print(out[-1], flush=True)
_lib.set_option("ablate", 0)
for rv, what in ((0, "6 lines x 6"), (1, "12 lines x 3"), (2, "4 lines x 9"), (3, "3 lines x 12"), (4, "6 lines x 5"), (0, "6 lines x 6")):
    _lib.set_option("ring_variant", rv)
    ix.search(q, 50)
    best = 1e9
    for _ in range(3):
        ix.search(q, 50)
        c = ix.counters()
        best = min(best, c["scan_ms"] / c["n_passes"])
    per = c["algorithmic_bytes"] / c["n_passes"]
    out.append({"ring_variant": rv, "what": what, "scan_ms_per_pass": best, "GBps": per / best / 1e6})
    print(out[-1], flush=True)
_lib.set_option("ring_variant", 0)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "ablate.json"), "w"), indent=1)
