#!/usr/bin/env python
"""Small-M encoder GEMMs (a rerank batch: 32 pairs x ~180 tokens = 5 632 packed rows, d = 1024): the persistent 256 x 256 kernel
has fewer tiles than the chip has CUs there (88 for N = 1024).  Variants 7 (persistent 256 x 256), 5 (256 x 256 one tile per
block), 2 / 4 (256 x 128), 1 (128 x 128, two blocks per CU), interleaved rounds:  python profiles/gemm_small_m.py [M]"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bergen_amd import encoder
M = int(sys.argv[1]) if len(sys.argv) > 1 else 5632
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
shapes = [("qk", M, 2048, 1024, False), ("attn_out", M, 1024, 1024, False), ("ffn1", M, 4096, 1024, True), ("ffn2", M, 1024, 4096, False)]
variants = [7, 5, 2, 1]
for name, m, n, k, gelu in shapes:
    a = (torch.randn(m, k, generator=g, device=dev) * 0.5).half()
    w = (torch.randn(n, k, generator=g, device=dev) * 0.05).half()
    bias = (torch.randn(n, generator=g, device=dev) * 0.1).half()
    out = torch.empty(m, n, dtype=torch.float16, device=dev)
    t = {v: [] for v in variants}
    for rnd in range(5):
        for v in variants:
            try:
                _, ms = encoder.gemm_f16(a, w, bias=bias, bias_mode=1, gelu=gelu, variant=v, out=out, repeats=9)
                t[v].append(ms)
            except Exception as e:
                t[v].append(float("nan"))
    row = {"shape": name, "M": m, "N": n, "K": k, "tiles_256": (m // 256) * (n // 256)}
    for v in variants:
        s = sorted(t[v]); row[f"v{v}_ms"] = round(s[len(s) // 2], 4)
    print(json.dumps(row), flush=True)
