#!/usr/bin/env python
"""Parity + race screen of the experimental gemm_f16_p16a.h (gemm_mfma16 = 5) against gemm_f16_p16.h (= 1): same bits, repeated."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bergen_amd import _lib, encoder  # noqa: E402
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(9)
bad = runs = 0
for (m, n, k) in [(256, 256, 64), (512, 768, 128), (2048, 1024, 192), (33280, 1536, 768), (33280, 768, 3072), (8192, 3072, 64), (66560, 768, 768), (2560, 1024, 1024), (768, 25600, 128)]:
    a = (torch.randn(m, k, generator=g, device=dev) * 0.5).half()
    w = (torch.randn(n, k, generator=g, device=dev) * 0.05).half()
    bias = (torch.randn(n, generator=g, device=dev) * 0.1).half()
    for kw in (dict(bias=bias), dict(bias=bias, gelu=True)):
        _lib.set_option("gemm_mfma16", 1)
        ref = encoder.gemm_f16(a, w, variant=7, **kw)[0].clone()
        _lib.set_option("gemm_mfma16", 5)
        for r in range(reps):
            out = encoder.gemm_f16(a, w, variant=7, **kw)[0]
            runs += 1
            if not torch.equal(out, ref):
                bad += 1
                d = (out != ref)
                print(f"MISMATCH {m}x{n}x{k} {sorted(kw)} rep {r}: {d.sum().item()} elements, first rows {d.any(1).nonzero()[:4].flatten().tolist()} cols {d.any(0).nonzero()[:4].flatten().tolist()}", flush=True)
                break
_lib.set_option("gemm_mfma16", 1)
print(f"{runs} launches, {bad} shapes differ")
sys.exit(1 if bad else 0)
