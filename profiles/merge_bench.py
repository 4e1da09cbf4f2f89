"""bh_merge_topk_device alone: time per call (HIP events around the library call) for n_lists x nq x k, sorted lists (binary-search ranking)
and shuffled ones (all-pairs ranking).   python profiles/merge_bench.py"""
import os, sys, time, statistics
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bergen_amd.index import merge_topk
from bergen_amd import _lib
_lib.init(0)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
for n_lists, nq, k in ((8, 2837, 50), (8, 1000, 200), (2, 1000, 200), (4, 1000, 200), (8, 1000, 100), (8, 250, 200)):
    s = torch.randn((n_lists, nq, k), generator=g, device=dev).sort(dim=2, descending=True).values.contiguous()
    i = torch.randint(0, 20_000_000, (n_lists, nq, k), generator=g, device=dev)
    out = (torch.empty((nq, k), dtype=torch.float32, device=dev), torch.empty((nq, k), dtype=torch.int64, device=dev))
    for label, (ss, ii) in (("sorted", (s, i)), ("shuffled", (s.flip(2).contiguous(), i))):
        merge_topk(ss, ii, out=out)
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            merge_topk(ss, ii, out=out)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print(f"{n_lists} lists x {nq} queries x k={k} {label}: {statistics.median(ts):.3f} ms (min {min(ts):.3f})", flush=True)
