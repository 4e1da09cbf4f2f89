#!/usr/bin/env python
"""Configuration sweep + ablation timings (ABL variants 11-16 / 21-26) of the encoder GEMM on the GPU box."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bergen_amd import encoder  # noqa: E402

WHAT = {1: "cfg1 BK64 128x128 4w r2 x2/CU", 2: "cfg2 BK32 256x128 4w r3 x2/CU", 3: "cfg3 BK32 256x256 8w r4",
        4: "cfg4 BK64 256x128 8w r3", 5: "cfg5 BK64 256x256 8w r2", 6: "generic (bounds-checked)"}
for base, nm in ((10, "cfg5"), (20, "cfg2")):
    for i, w in enumerate(["no LDS-DMA in loop", "DMA + barriers only", "no fragment reads", "no epilogue",
                           "MFMA + barriers only", "MFMA + barriers, no epilogue"], 1):
        WHAT[base + i] = f"{nm}: {w}"
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 3, 4, 5, 21, 22, 23, 24, 25, 26, 14, 16]
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
out = []
for name, m, n, k, gelu in [("ffn1", 66560, 3072, 768, True), ("qk", 66560, 1536, 768, False), ("ffn2", 66560, 768, 3072, False),
                            ("big", 8192, 8192, 8192, False)]:
    a = (torch.randn(m, k, generator=g, device=dev) * 0.5).half()
    w = (torch.randn(n, k, generator=g, device=dev) * 0.05).half()
    bias = (torch.randn(n, generator=g, device=dev) * 0.1).half()
    c = torch.empty(m, n, dtype=torch.float16, device=dev)
    for v in variants:
        _, ms = encoder.gemm_f16(a, w, bias=bias, gelu=gelu, variant=v, out=c, repeats=11)
        tf = 2.0 * m * n * k / (ms * 1e-3) / 1e12
        out.append({"shape": name, "variant": v, "what": WHAT.get(v, ""), "ms": ms, "tflops_equiv": tf})
        print(f"{name:5s} v{v:2d} {WHAT.get(v, ''):32s} {ms:8.3f} ms {tf:8.1f} TF/s-equivalent", flush=True)
    del a, w, c
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "gemm_ablation.json"), "w"), indent=1)
