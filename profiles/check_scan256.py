"""One-shot check of the 256-query scan kernel (option scan_kernel 3, scan_topk256.hip) against the 4-wave kernel
(scan_kernel 0): bit-equality on small shapes (dims 384 / 512 / 768, candidate lists 64 / 128 / 256, pass boundaries,
exact ties), then queries/s on the headline workload for both kernels, the ring / pipeline variants and the in-kernel
ablations.  Run on the GPU box:  python profiles/check_scan256.py [--no-headline] [--variants]"""
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
        _lib.set_option("scan_kernel", 3)


def main():
    _lib.init(0)
    out = {"small": [], "headline": []}
    rng = np.random.default_rng(1)
    shapes = ((33, 768, 1, 5), (9001, 768, 255, 50), (9001, 768, 256, 50), (9001, 768, 257, 50), (70001, 768, 600, 50),
              (5000, 768, 70, 50), (40001, 768, 300, 100), (40001, 768, 260, 200), (30001, 384, 300, 50),
              (30001, 512, 300, 50), (30001, 500, 100, 120), (300001, 768, 64, 50))
    for n, d, nq, k in shapes:
        x = rng.standard_normal((n, d)).astype(np.float16)
        if n == 5000:
            x = rng.integers(-2, 3, size=(n, d)).astype(np.float16)  # heavy exact ties
        q = rng.standard_normal((nq, d)).astype(np.float16)
        ix = bergen_amd.FlatIndex(n, d, metric="ip")
        ix.upload(x)
        ix.finalize()
        s0, i0 = search(ix, q, k, 0)
        t0 = ix.counters()["query_tile"]
        s3, i3 = search(ix, q, k, 3)
        same = bool(np.array_equal(np.asarray(i0), np.asarray(i3)) and
                    np.array_equal(np.asarray(s0).view(np.uint32), np.asarray(s3).view(np.uint32)))
        out["small"].append({"n": n, "d": d, "nq": nq, "k": k, "same": same, "tile0": t0, "tile3": ix.counters()["query_tile"]})
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
        runs = [(0, 0, 0), (3, 0, 0), (2, 0, 0), (3, 0, 0)]
        if "--variants" in sys.argv:
            runs += [(3, rv, 0) for rv in (1, 2, 3, 4)] + [(3, 0, ab) for ab in (1, 3, 7, 9, 11)] + [(3, 0, 0)]
        for kern, rv, ab in runs:
            _lib.set_option("scan_kernel", kern)
            _lib.set_option("ring_variant", rv)
            _lib.set_option("ablate", ab)
            s, i = ix.search(q, k)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s, i = ix.search(q, k)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            c = ix.counters()
            if base is None:
                base = (s.clone(), i.clone())
            elif not (torch.equal(s, base[0]) and torch.equal(i, base[1])) and ab == 0:
                badq = torch.nonzero((i != base[1]).any(dim=1) | (s != base[0]).any(dim=1)).flatten()
                msg = {"mismatching_queries": int(badq.numel()), "first": []}
                for qi in badq[:6].tolist():
                    r = int(torch.nonzero((i[qi] != base[1][qi]) | (s[qi] != base[0][qi])).flatten()[0])
                    msg["first"].append({"q": qi, "rank": r, "got": [int(i[qi, r]), float(s[qi, r])], "want": [int(base[1][qi, r]), float(base[0][qi, r])],
                                         "want_id_in_got": bool((i[qi] == base[1][qi, r]).any())})
                print(msg, file=sys.stderr, flush=True)
            out["headline"].append({"scan_kernel": kern, "ring_variant": rv, "ablate": ab, "qps": nq / dt, "tile": c["query_tile"],
                                    "passes": c["n_passes"], "scan_ms_per_pass": c["scan_ms"] / c["n_passes"], "shader_mhz": c["shader_mhz"],
                                    "same_as_kernel0": bool(torch.equal(s, base[0]) and torch.equal(i, base[1]))})
            print(out["headline"][-1], file=sys.stderr, flush=True)
        _lib.set_option("scan_kernel", 3)
        _lib.set_option("ring_variant", 0)
        _lib.set_option("ablate", 0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
