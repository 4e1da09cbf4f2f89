"""What bounds the 8-GPU strong-scaling case, measured on ONE GPU: the headline search on a shard of N / g rows for
g = 1, 2, 4, 8 (one rank's work when the corpus is row-sharded over g GPUs), with the per-search fixed costs broken out.
Run on the GPU box:  python profiles/shard_sweep.py > profiles/r02_shard_sweep.json"""
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
    _lib.init(0)
    # python profiles/shard_sweep.py [dim k queries [g ...]]   (defaults: the headline geometry; configs[4]: 1024 200 1000)
    for kv in [x for x in sys.argv[1:] if "=" in x]:  # library options, e.g. ring_variant=5
        _lib.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    av = [int(x) for x in sys.argv[1:] if "=" not in x]
    dim, k, nq = (av + [768, 50, 2837][len(av):])[:3] if len(av) < 3 else av[:3]
    n_total = 21_000_000
    shards = tuple(av[3:]) or (1, 2, 4, 8)
    dev = torch.device("cuda", 0)
    q = bench.make_queries(nq, dim, dev)
    out = {"workload": f"{nq} queries x (21 M / g) x {dim} fp16, top-{k}, one GPU", "shards": []}
    for g in shards:
        lo, hi = bergen_amd.shard_range(n_total, 0, g)
        ix = bergen_amd.FlatIndex(hi - lo, dim, metric="ip", device=0)
        bench.fill_shard(ix, lo, hi, dim, q, n_total, dev)
        ix.finalize()
        ix.search(q, k)
        torch.cuda.synchronize()
        reps = 5
        t0 = time.perf_counter()
        scan = merge = total = t_search = t_d2h = 0.0
        for _ in range(reps):
            ta = time.perf_counter()
            s, i = ix.search(q, k)
            tb = time.perf_counter()
            host = (s.cpu(), i.cpu())
            t_d2h += time.perf_counter() - tb
            t_search += tb - ta
            c = ix.counters()
            scan += c["scan_ms"]
            merge += c["merge_ms"]
            total += c["total_ms"]
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps * 1e3
        out["shards"].append({"g": g, "rows": hi - lo, "queries_per_s": nq / (wall * 1e-3), "wall_ms": wall,
                              "stream_total_ms": total / reps, "scan_ms": scan / reps, "merge_ms_side_stream": merge / reps,
                              "passes": c["n_passes"], "scan_ms_per_pass": scan / reps / c["n_passes"],
                              "non_scan_ms": wall - scan / reps, "search_call_ms": t_search / reps * 1e3, "result_d2h_ms": t_d2h / reps * 1e3,
                              "query_tile": c["query_tile"]})
        print(out["shards"][-1], file=sys.stderr, flush=True)
        ix.close()
    base = out["shards"][0]["queries_per_s"]
    for sh in out["shards"]:
        sh["speedup_vs_full_corpus"] = sh["queries_per_s"] / base
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
