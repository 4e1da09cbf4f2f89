"""Paired-workgroup scan (query_split 2) tuning on the headline workload: one index build, then a sweep over
pair_window x nontemporal (and the unsplit kernel as the reference).  Run on the GPU box from the repo root:
    python profiles/bench_split.py [n_rows] > gpurun_out/bench_split.json"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import bergen_amd  # noqa: E402
from bergen_amd import _lib  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 21_000_000
    dim, k, nq = 768, 50, 2837
    _lib.init(0)
    dev = torch.device("cuda", 0)
    q = bench.make_queries(nq, dim, dev)
    ix = bergen_amd.FlatIndex(n, dim, metric="ip", device=0)
    bench.fill_shard(ix, 0, n, dim, q, n, dev)
    ix.finalize()
    res = []
    base = None
    ablate = int(os.environ.get("ABLATE", "0"))  # 2 = stream only, 3 = + LDS reads, 4 = MFMA only (timings only)
    _lib.set_option("ablate", ablate)
    configs = [(0, split, il, nt) for il in (0, 1) for split in (1, 2) for nt in (1, 0)]
    for kern, split, window, nt in configs:  # (window slot reused: dma_interleave)
        _lib.set_option("scan_kernel", kern)
        _lib.set_option("dma_interleave", window)
        _lib.set_option("query_split", split)
        _lib.set_option("nontemporal", nt)
        s, i = ix.search(q, k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            s, i = ix.search(q, k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 2
        c = ix.counters()
        if base is None:
            base = (s.clone(), i.clone())
        same = bool(torch.equal(s, base[0]) and torch.equal(i, base[1]))
        res.append({"scan_kernel": kern, "query_split": split, "dma_interleave": window, "nt": nt, "qps": nq / dt, "passes": c["n_passes"],
                    "scan_ms_per_pass": c["scan_ms"] / c["n_passes"], "same_results_as_unsplit": same, "ablate": ablate})
        print(res[-1], file=sys.stderr, flush=True)
    _lib.set_option("ablate", 0)
    _lib.set_option("scan_kernel", 0)
    _lib.set_option("dma_interleave", 0)
    _lib.set_option("query_split", 1)
    _lib.set_option("nontemporal", 1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
