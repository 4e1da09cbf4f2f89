"""One search on a shard of N / g rows for a rocprofv3 kernel trace (where do the microseconds between the scan launches go):
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_g8 -o t -- python profiles/shard_trace.py 8"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import bergen_amd  # noqa: E402
from bergen_amd import _lib  # noqa: E402

g = int(sys.argv[1]) if len(sys.argv) > 1 else 8
_lib.init(0)
dim, k, nq, n_total = 768, 50, 2837, 21_000_000
dev = torch.device("cuda", 0)
q = bench.make_queries(nq, dim, dev)
lo, hi = bergen_amd.shard_range(n_total, 0, g)
ix = bergen_amd.FlatIndex(hi - lo, dim, metric="ip", device=0)
bench.fill_shard(ix, lo, hi, dim, q, n_total, dev)
ix.finalize()
for _ in range(3):
    s, i = ix.search(q, k)
    host = (s.cpu(), i.cpu())
torch.cuda.synchronize()
