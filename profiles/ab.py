"""A/B of scan options inside ONE process (box-to-box and run-to-run clock differences are larger than most effects):
python profiles/ab.py <g> <option> <v0> <v1> [rounds]  ->  median scan ms per pass of each value, alternating."""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import bergen_amd  # noqa: E402
from bergen_amd import _lib  # noqa: E402


def main():
    g, opt, v0, v1 = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 6
    _lib.init(0)
    dim, k, nq, n_total = 768, 50, 2837, 21_000_000
    dev = torch.device("cuda", 0)
    q = bench.make_queries(nq, dim, dev)
    lo, hi = bergen_amd.shard_range(n_total, 0, g)
    ix = bergen_amd.FlatIndex(hi - lo, dim, metric="ip", device=0)
    bench.fill_shard(ix, lo, hi, dim, q, n_total, dev)
    ix.finalize()
    res = {v0: [], v1: []}
    tot = {v0: [], v1: []}
    for r in range(rounds + 1):
        for v in (v0, v1):
            _lib.set_option(opt, v)
            ix.search(q, k)
            c = ix.counters()
            if r:  # round 0 warms up
                res[v].append(c["scan_ms"] / c["n_passes"])
                tot[v].append(c["total_ms"])
    for v in (v0, v1):
        print(f"g={g} {opt}={v}: scan ms/pass median {statistics.median(res[v]):.4f} (min {min(res[v]):.4f} max {max(res[v]):.4f}); "
              f"search ms median {statistics.median(tot[v]):.3f}")


if __name__ == "__main__":
    main()
