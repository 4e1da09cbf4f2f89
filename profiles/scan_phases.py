"""Where a scan launch of the 8-GPU shard size spends its time: per-workgroup phase stamps (100 MHz) and cold-path counts
of the LAST launch of a search, from the kernel's diagnostics block.  python profiles/scan_phases.py [g]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import bergen_amd  # noqa: E402
from bergen_amd import _lib  # noqa: E402

TL_WORDS = 8 * 6 * 2 * 5


def main():
    g = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    nq = int(sys.argv[2]) if len(sys.argv) > 2 else 2816
    _lib.init(0)
    for kv in sys.argv[3:]:
        k_, v_ = kv.split('=')
        _lib.set_option(k_, int(v_))
        print('option', k_, v_)
    dim, k, n_total = int(os.environ.get("BH_DIM", "768")), int(os.environ.get("BH_K", "50")), 21_000_000  # (configs[4]: BH_DIM=1024 BH_K=200)
    dev = torch.device("cuda", 0)
    q = bench.make_queries(nq, dim, dev)
    lo, hi = bergen_amd.shard_range(n_total, 0, g)
    ix = bergen_amd.FlatIndex(hi - lo, dim, metric="ip", device=0)
    bench.fill_shard(ix, lo, hi, dim, q, n_total, dev)
    ix.finalize()
    for _ in range(2):
        ix.search(q, k)
    c = ix.counters()
    grid = c["n_workgroups"]
    buf = (ctypes.c_uint64 * (grid * 8))()
    n = _lib.lib().bh_debug_scan_timeline(ix._h, buf, len(buf))
    a = np.frombuffer(buf, dtype=np.uint64)[:n].astype(np.int64)
    w = a[:grid * 8].reshape(grid, 8)
    # entry, loop start, loop end, exit (100 MHz ticks); candidates wave 0 holds
    st = np.stack([w[:, 0], w[:, 1], w[:, 2], w[:, 3], w[:, 6]], axis=1)
    t0 = st[:, 0].min()
    us = lambda x: x * 0.01
    print(f"g={g} scan_ms_per_pass={c['scan_ms'] / c['n_passes']:.4f} mhz={c['shader_mhz']:.0f}")
    print(f"entry skew        : max {us(st[:, 0].max() - t0):8.1f} us")
    print(f"prologue          : mean {us((st[:, 1] - st[:, 0]).mean()):8.1f} us")
    print(f"tile loop         : mean {us((st[:, 2] - st[:, 1]).mean()):8.1f}  min {us((st[:, 2] - st[:, 1]).min()):8.1f}  max {us((st[:, 2] - st[:, 1]).max()):8.1f} us")
    print(f"final sort+publish: mean {us((st[:, 3] - st[:, 2]).mean()):8.1f} us")
    print(f"first entry -> last end: {us(st[:, 3].max() - t0):8.1f} us; loop end spread {us(st[:, 2].max() - st[:, 2].min()):8.1f} us")
    loop = us(st[:, 2] - st[:, 1])
    print("tile loop by XCD (b % 8): " + " ".join(f"{loop[x::8].mean():7.1f}" for x in range(8)))
    print("   spread inside an XCD : " + " ".join(f"{loop[x::8].max() - loop[x::8].min():7.1f}" for x in range(8)))
    order = np.argsort(loop)
    print("slowest workgroups:", [(int(b), round(float(loop[b]), 1)) for b in order[-6:]])
    print("fastest workgroups:", [(int(b), round(float(loop[b]), 1)) for b in order[:6]])
    print(f"wave 0: candidates held at the end (32 queries) {st[:, 4].mean():.1f}")


if __name__ == "__main__":
    main()
