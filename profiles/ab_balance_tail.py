"""Same-box A/B of option balance_tail (index.hip: the remainder of a query set as one balanced paired launch with idle waves, instead
of an unpaired full pass + a 128-query tail pass) at the headline geometry:
    python profiles/ab_balance_tail.py [g [dim [k [nq]]]]     (g = 1: all 21 M rows; g = 8: one of eight shards)
Interleaved rounds of balance_tail = 0 / 1; one JSON line per setting: whole-search ms (median of the library's own events), the
remainder's part (balanced launch, or unpaired pass + tail pass), queries/s, identity of all results with balance_tail = 0."""
import json, os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, bergen_amd
from bergen_amd import _lib
g = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
k = int(sys.argv[3]) if len(sys.argv) > 3 else 50
nq = int(sys.argv[4]) if len(sys.argv) > 4 else (2837 if dim == 768 else 1000)
_lib.init(0)
dev = torch.device("cuda", 0)
q = bench.make_queries(nq, dim, dev)
lo, hi = bergen_amd.shard_range(21_000_000, 0, g)
ix = bergen_amd.FlatIndex(hi - lo, dim, metric="ip", device=0)
bench.fill_shard(ix, lo, hi, dim, q, 21_000_000, dev)
ix.finalize()
times, rest, ref, ok = {0: [], 1: []}, {0: [], 1: []}, None, {}
for rnd in range(8):
    for on in (0, 1):
        ix.set_option("balance_tail", on)
        s, i = ix.search(q, k)
        c = ix.counters()
        if rnd:
            times[on].append(c["total_ms"])
            n_full = c["paired_launches"] - (1 if c["balanced_queries"] else 0)
            full_ms = (c["paired_scan_ms"] - c["balanced_scan_ms"])
            rest[on].append(c["scan_ms"] - full_ms)   # everything that is not a full paired launch: the remainder's launches
        if ref is None:
            ref = (s.clone(), i.clone())
        ok[on] = bool(torch.equal(i, ref[1]) and torch.equal(s, ref[0]))
for on in (0, 1):
    t = statistics.median(times[on])
    print(json.dumps({"g": g, "rows": hi - lo, "dim": dim, "k": k, "queries": nq, "balance_tail": on, "search_ms_median": round(t, 3),
                      "search_ms_min": round(min(times[on]), 3), "remainder_launches_ms_median": round(statistics.median(rest[on]), 3),
                      "queries_per_s": round(nq / t * 1e3, 1), "identical_to_balance_tail_0": ok[on]}), flush=True)
