#!/usr/bin/env python
"""In-kernel ablations of the 16x16x32 persistent GEMM (gemm_f16_p16.h template parameter ABL; results of the ablated runs are invalid) on
the FFN-up shape (bias + GELU, M = 65 536, N = 3072, K = 768: 3072 tiles = 12 whole rounds) and the FFN-down K:
  python profiles/gemm_p16_ablate.py [out.json]
ABL bits: 1 no LDS-DMA after the pipeline start, 2 no MFMA, 4 no fragment reads, 8 no epilogue, 16 epilogue math without stores."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bergen_amd import _lib, encoder  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "gemm_p16_ablate.json")
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
NAMES = {0: "production", 8: "no epilogue", 16: "epilogue math, no stores", 9: "MFMA + fragment reads + barriers (no DMA, no epilogue)",
         10: "DMA + fragment reads + barriers (no MFMA, no epilogue)", 12: "DMA + MFMA + barriers (no fragment reads, no epilogue)",
         14: "DMA + barriers only", 13: "MFMA + barriers only", 11: "fragment reads + barriers only", 1: "no DMA", 2: "no MFMA", 4: "no fragment reads"}
res = []
for (m, n, k) in [(65536, 3072, 768), (65536, 768, 3072)]:
    a = (torch.randn(m, k, generator=g, device=dev) * 0.5).half()
    w = (torch.randn(n, k, generator=g, device=dev) * 0.05).half()
    bias = (torch.randn(n, generator=g, device=dev) * 0.1).half()
    outb = torch.empty(m, n, dtype=torch.float16, device=dev)
    times = {}
    for rnd in range(3):
        for abl in NAMES:
            _lib.set_option("gemm_mfma16", 16 * abl + 1 if abl else 1)
            _, ms = encoder.gemm_f16(a, w, bias=bias, bias_mode=1, gelu=True, out=outb, repeats=7)
            times.setdefault(abl, []).append(ms)
    for abl, t in times.items():
        t = sorted(t)[len(t) // 2]
        row = {"M": m, "N": n, "K": k, "abl": abl, "what": NAMES[abl], "ms": round(t, 4), "tflops_equivalent": round(2.0 * m * n * k / (t * 1e-3) / 1e12, 1)}
        res.append(row)
        print(json.dumps(row), flush=True)
    del a, w, outb
_lib.set_option("gemm_mfma16", 1)  # (the library default)
json.dump(res, open(out, "w"), indent=1)
