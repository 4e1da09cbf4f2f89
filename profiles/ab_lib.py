"""A/B of two BUILDS of the library on one box (box-to-box clock differences are larger than most effects): run this once per
build, selecting the other one with BERGEN_HIP_LIB=<path to its libbergen_hip.so>:
    python profiles/ab_lib.py <g> <k>      (BH_DIM=1024 for the configs[4] geometry)
prints the scan time per full pass (median of 8 searches) of a shard of 21 M / g rows for this process's library."""
import os, sys, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, bergen_amd
from bergen_amd import _lib
g, k = int(sys.argv[1]), int(sys.argv[2])
dim = int(os.environ.get("BH_DIM", "768"))
_lib.init(0)
dev = torch.device("cuda", 0)
nq = 2837 if dim == 768 else 1000
q = bench.make_queries(nq, dim, dev)
lo, hi = bergen_amd.shard_range(21_000_000, 0, g)
ix = bergen_amd.FlatIndex(hi - lo, dim, metric="ip", device=0)
bench.fill_shard(ix, lo, hi, dim, q, 21_000_000, dev)
ix.finalize()
ms, tot = [], []
for r in range(9):
    ix.search(q, k)
    c = ix.counters()
    if r:
        ms.append((c["scan_ms"] - c.get("tail_scan_ms", 0.0)) / max(1, c["n_passes"] - (1 if c.get("tail_scan_ms", 0) > 0 else 0)))
        tot.append(c["total_ms"])
print(f"{os.path.basename(os.path.dirname(_lib.LIB_PATH))}: g={g} d={dim} k={k}: main-pass ms median {statistics.median(ms):.4f} min {min(ms):.4f}; search ms median {statistics.median(tot):.3f}; mhz {c['shader_mhz']:.0f}")
