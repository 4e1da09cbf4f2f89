"""Per-wave stage timeline of the 256-query scan kernel (option ablate 5: s_memtime stamps of workgroup 0).
Run on the GPU box:  python profiles/scan_timeline.py [ring_variant]   -> JSON on stdout, a readable table on stderr."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import bergen_amd  # noqa: E402
from bergen_amd import _lib  # noqa: E402

TL_TILES, S = 6, 2


def main():
    rv = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    abl = int(sys.argv[2]) if len(sys.argv) > 2 else 32  # 32 = production + stamps; 32 | ablation bits
    _lib.init(0)
    n, dim, k, nq = 21_000_000, 768, 50, 256
    dev = torch.device("cuda", 0)
    q = bench.make_queries(nq, dim, dev)
    ix = bergen_amd.FlatIndex(n, dim, metric="ip", device=0)
    bench.fill_shard(ix, 0, n, dim, q, n, dev)
    ix.finalize()
    _lib.set_option("scan_kernel", 3)
    _lib.set_option("ring_variant", rv)
    _lib.set_option("ablate", abl)
    ix.search(q, k)
    ix.search(q, k)
    c = ix.counters()
    grid = c["n_workgroups"]
    words = grid * 8 + 8 * TL_TILES * S * 5
    buf = (ctypes.c_uint64 * words)()
    got = _lib.lib().bh_debug_scan_timeline(ix._h, buf, words)
    assert got == words, got
    a = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
    tl = a[grid * 8:].reshape(8, TL_TILES * S, 5)
    t_ref = tl[:, 0, 0].min()
    out = {"ring_variant": rv, "ablate": abl, "scan_ms": c["scan_ms"], "shader_mhz": c["shader_mhz"], "waves": []}
    print(f"scan {c['scan_ms']:.3f} ms, {c['shader_mhz']:.0f} MHz; per wave and stage: start | vmcnt wait | barrier wait | body | dma issue",
          file=sys.stderr)
    for w in range(8):
        rows = []
        for st in range(TL_TILES * S):
            t0, t1, t2, t3, dma = (int(x) for x in tl[w, st])
            rows.append({"start": t0 - int(t_ref), "vmcnt_wait": t1 - t0, "barrier_wait": t2 - t1, "body": t3 - t2, "dma_issue": dma})
        out["waves"].append(rows)
        print(f"wave {w}: " + "  ".join(f"{r['start']:6d}|{r['vmcnt_wait']:4d}|{r['barrier_wait']:4d}|{r['body']:5d}|{r['dma_issue']:4d}" for r in rows),
              file=sys.stderr)
    per_stage = (tl[:, -1, 3].max() - tl[:, 0, 0].min()) / (TL_TILES * S)
    out["cycles_per_stage"] = float(per_stage)
    print(f"cycles per stage {per_stage:.0f} (MFMA floor 1536)", file=sys.stderr)
    _lib.set_option("ablate", 0)
    _lib.set_option("ring_variant", 0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
