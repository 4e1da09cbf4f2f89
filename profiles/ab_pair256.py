"""Same-box A/B of option pair256 (two 256-query passes per launch on partner workgroups of one XCD, scan_topk256.hip) at the
headline geometry:  python profiles/ab_pair256.py [g [dim [k]]]   (g = 1: all 21 M rows; g = 8: one of eight shards; 1024 200 = configs[4])
Interleaved rounds of  pair256 = 0 / 1 (paced) / 2 (free-running), each with the non-temporal stream policy on and off;
prints one JSON line per configuration: whole-search ms (median), queries/s, and a 64-query spot check of ids against the
unpaired result."""
import json, os, statistics, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, bergen_amd
from bergen_amd import _lib
g = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
k = int(sys.argv[3]) if len(sys.argv) > 3 else 50
nq = 2837 if dim == 768 else 1000
_lib.init(0)
dev = torch.device("cuda", 0)
q = bench.make_queries(nq, dim, dev)
lo, hi = bergen_amd.shard_range(21_000_000, 0, g)
ix = bergen_amd.FlatIndex(hi - lo, dim, metric="ip", device=0)
bench.fill_shard(ix, lo, hi, dim, q, 21_000_000, dev)
ix.finalize()
configs = [(0, 1), (1, 1), (2, 1), (1, 0), (0, 0)]
if os.environ.get("AB_CONFIGS"):  # e.g. "0,1;1,1"
    configs = [tuple(int(v) for v in c.split(",")) for c in os.environ["AB_CONFIGS"].split(";")]
times = {c: [] for c in configs}
ref = None
ok = {}
for rnd in range(6):
    for c in configs:
        ix.set_option("pair256", c[0])
        ix.set_option("nontemporal", c[1])
        s, i = ix.search(q, k)
        cnt = ix.counters()
        if rnd:
            times[c].append(cnt["total_ms"])
        if ref is None:
            ref = (s.clone(), i.clone())
        ok[c] = bool(torch.equal(i, ref[1]) and torch.equal(s, ref[0]))
for c in configs:
    t = statistics.median(times[c])
    print(json.dumps({"lib": os.path.basename(os.path.dirname(_lib.LIB_PATH)), "g": g, "rows": hi - lo, "dim": dim, "k": k, "pair256": c[0], "nontemporal": c[1], "search_ms_median": round(t, 3), "search_ms_min": round(min(times[c]), 3),
                      "queries_per_s": round(nq / t * 1e3, 1), "identical_to_unpaired": ok[c]}), flush=True)
