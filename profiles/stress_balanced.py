"""Stress of the balanced remainder / idle waves (index.hip balance_tail, scan_topk256.hip nq_valid): the same searches repeated many
times on the full corpus and on an eighth of it must return the SAME bits every time (a pacing or claim race would show up as a rare
different list — or as a hang, which the caller's timeout turns into a failure).  python profiles/stress_balanced.py [iterations]"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, bergen_amd
from bergen_amd import _lib
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
_lib.init(0)
dev = torch.device("cuda", 0)
out = []
for g, dim, k in ((8, 768, 50), (1, 768, 50), (8, 1024, 200)):
    q_all = bench.make_queries(2837, dim, dev)
    lo, hi = bergen_amd.shard_range(21_000_000, 0, g)
    ix = bergen_amd.FlatIndex(hi - lo, dim, metric="ip", device=0)
    bench.fill_shard(ix, lo, hi, dim, q_all, 21_000_000, dev)
    ix.finalize()
    for nq in ((277, 2837, 300, 789, 511, 33) if dim == 768 else (200, 456, 1000)):
        q = q_all[:nq]
        ref, bad, t0 = None, 0, time.perf_counter()
        n_it = iters if g == 8 else max(8, iters // 4)
        for it in range(n_it):
            s, i = ix.search(q, k)
            if ref is None:
                ref = (s.clone(), i.clone())
                c = ix.counters()
            elif not (torch.equal(i, ref[1]) and torch.equal(s, ref[0])):
                bad += 1
        out.append({"g": g, "dim": dim, "k": k, "queries": nq, "iterations": n_it, "different_results": bad, "balanced_queries": c["balanced_queries"],
                    "seconds": round(time.perf_counter() - t0, 2)})
        print(json.dumps(out[-1]), flush=True)
    ix.close()
assert all(o["different_results"] == 0 for o in out)
print("stress OK")
