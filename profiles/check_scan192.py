"""One-shot check of the 192-query scan kernel (option scan_kernel 2) against the production kernel: bit-equality on small
shapes, then queries/s on the headline workload.  Run on the GPU box: python profiles/check_scan192.py"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import bergen_amd  # noqa: E402
from bergen_amd import _lib  # noqa: E402


def search(ix, q, k, kern):
    _lib.set_option("scan_kernel", kern)
    try:
        return ix.search(q, k)
    finally:
        _lib.set_option("scan_kernel", 0)


def main():
    _lib.init(0)
    out = {"small": [], "headline": []}
    rng = np.random.default_rng(1)
    for n, nq, k in ((33, 1, 5), (9001, 191, 50), (9001, 192, 50), (9001, 193, 50), (70001, 400, 50), (5000, 70, 50)):
        x = rng.standard_normal((n, 768)).astype(np.float16)
        if n == 5000:
            x = rng.integers(-2, 3, size=(n, 768)).astype(np.float16)  # heavy exact ties
        q = rng.standard_normal((nq, 768)).astype(np.float16)
        ix = bergen_amd.FlatIndex(n, 768, metric="ip")
        ix.upload(x)
        ix.finalize()
        s0, i0 = search(ix, q, k, 0)
        s2, i2 = search(ix, q, k, 2)
        same = bool(np.array_equal(np.asarray(i0), np.asarray(i2)) and
                    np.array_equal(np.asarray(s0).view(np.uint32), np.asarray(s2).view(np.uint32)))
        out["small"].append({"n": n, "nq": nq, "k": k, "same": same, "tile": ix.counters()["query_tile"]})
        ix.close()
        print(out["small"][-1], file=sys.stderr, flush=True)
    if all(r["same"] for r in out["small"]) and "--no-headline" not in sys.argv:
        n, dim, k, nq = 21_000_000, 768, 50, 2837
        dev = torch.device("cuda", 0)
        q = bench.make_queries(nq, dim, dev)
        ix = bergen_amd.FlatIndex(n, dim, metric="ip", device=0)
        bench.fill_shard(ix, 0, n, dim, q, n, dev)
        ix.finalize()
        base = None
        for kern in (0, 2, 0, 2):
            _lib.set_option("scan_kernel", kern)
            s, i = ix.search(q, k)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s, i = ix.search(q, k)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            c = ix.counters()
            if base is None:
                base = (s.clone(), i.clone())
            out["headline"].append({"scan_kernel": kern, "qps": nq / dt, "passes": c["n_passes"], "scan_ms_per_pass": c["scan_ms"] / c["n_passes"],
                                    "same_as_kernel0": bool(torch.equal(s, base[0]) and torch.equal(i, base[1]))})
            print(out["headline"][-1], file=sys.stderr, flush=True)
        _lib.set_option("scan_kernel", 0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
