#!/usr/bin/env python
"""Race screen of the 16x16x32 persistent GEMM (gemm_f16_p16.h): every epilogue it has, on shapes with whole and partial rounds of tiles, K from
one stage to 48, repeated; every repetition must give the bits of the first, and the bits of the 32x32x16 kernel (gemm_mfma16 = 0).  With and
without the tail split.   python profiles/stress_gemm_p16.py [repeats]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bergen_amd import _lib, encoder  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(5)
bad = 0
runs = 0
for (m, n, k) in [(33280, 1536, 768), (33280, 768, 3072), (8192, 3072, 64), (66560, 768, 768), (2560, 1024, 1024), (256, 256, 192)]:
    a = (torch.randn(m, k, generator=g, device=dev) * 0.5).half()
    w = (torch.randn(n, k, generator=g, device=dev) * 0.05).half()
    bias = (torch.randn(n, generator=g, device=dev) * 0.1).half()
    for kw in (dict(), dict(bias=bias), dict(bias=bias, gelu=True), dict(bias=bias, gelu="swiglu")):
        _lib.set_option("gemm_mfma16", 0)
        ref, _ = encoder.gemm_f16(a, w, variant=7, **kw)
        ref = ref.clone()
        for split in (0, 1):
            _lib.set_option("gemm_mfma16", 1)
            _lib.set_option("gemm_tail_split", split)
            for r in range(reps):
                out, _ = encoder.gemm_f16(a, w, variant=7, **kw)
                runs += 1
                if not torch.equal(out, ref):
                    bad += 1
                    print(f"MISMATCH {m}x{n}x{k} {sorted(kw)} tail_split={split} repetition {r}: {(out != ref).sum().item()} elements", flush=True)
_lib.set_option("gemm_tail_split", 0)
print(f"{runs} launches of the 16x16x32 kernel over 6 shapes x 4 epilogues x tail split off / on: {bad} differ from the 32x32x16 kernel's bits")
sys.exit(1 if bad else 0)
