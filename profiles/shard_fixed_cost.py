"""Per-pass fixed cost of the scan kernel at the 8-GPU shard size (N / 8 rows on one GPU): the same search with parts of
the kernel switched off (bench-only ablation flags; results invalid by design) beside the production kernel.
Run on the GPU box:  python profiles/shard_fixed_cost.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import bergen_amd  # noqa: E402
from bergen_amd import _lib  # noqa: E402


def main():
    _lib.init(0)
    dim, k, nq, n_total = 768, 50, 2837, 21_000_000
    dev = torch.device("cuda", 0)
    q = bench.make_queries(nq, dim, dev)
    out = []
    for g in (8, 1):
        lo, hi = bergen_amd.shard_range(n_total, 0, g)
        ix = bergen_amd.FlatIndex(hi - lo, dim, metric="ip", device=0)
        bench.fill_shard(ix, lo, hi, dim, q, n_total, dev)
        ix.finalize()
        for name, opts in (("production", {}), ("no_share", {"share_threshold": 0}), ("no_filter", {"ablate": 1}),
                           ("stream_only", {"ablate": 7})):
            for o, v in opts.items():
                _lib.set_option(o, v)
            _lib.set_option("certify", 0 if opts else 1)
            ix.search(q, k)
            scan = 0.0
            for _ in range(3):
                ix.search(q, k)
                c = ix.counters()
                scan += c["scan_ms"] / c["n_passes"] / 3
            out.append({"g": g, "variant": name, "scan_ms_per_pass": scan, "shader_mhz": c["shader_mhz"]})
            print(out[-1], file=sys.stderr, flush=True)
            for o in opts:
                _lib.set_option(o, 1 if o == "share_threshold" else 0)
            _lib.set_option("certify", 1)
        ix.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
