import os, sys, time, statistics
import torch
sys.path.insert(0, os.getcwd())
from bergen_amd.index import merge_topk
from bergen_amd import _lib
_lib.init(0)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
for n_lists, nq, k in ((8, 256, 200), (8, 257, 200), (8, 512, 200), (8, 513, 200), (8, 768, 200), (8, 1000, 200), (8, 1000, 160), (8, 1000, 128), (8, 1000, 120), (8,2000,200)):
    s = torch.randn((n_lists, nq, k), generator=g, device=dev).sort(dim=2, descending=True).values.contiguous()
    i = torch.randint(0, 20_000_000, (n_lists, nq, k), generator=g, device=dev)
    out = (torch.empty((nq, k), dtype=torch.float32, device=dev), torch.empty((nq, k), dtype=torch.int64, device=dev))
    merge_topk(s, i, out=out); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); merge_topk(s, i, out=out); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{n_lists} x {nq} x k={k}: {statistics.median(ts):.3f} ms", flush=True)
