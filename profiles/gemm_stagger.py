#!/usr/bin/env python
"""Epilogue ablations + start-stagger sweep of the encoder GEMM (bench-only knobs) on the GPU box."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bergen_amd import _lib, encoder  # noqa: E402

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
out = []


def run(name, m, n, k, gelu, variants, tag=""):
    a = (torch.randn(m, k, generator=g, device=dev) * 0.5).half()
    w = (torch.randn(n, k, generator=g, device=dev) * 0.05).half()
    bias = (torch.randn(n, generator=g, device=dev) * 0.1).half()
    c = torch.empty(m, n, dtype=torch.float16, device=dev)
    for v in variants:
        _, ms = encoder.gemm_f16(a, w, bias=bias, gelu=gelu, variant=v, out=c, repeats=11)
        tf = 2.0 * m * n * k / (ms * 1e-3) / 1e12
        out.append({"shape": name, "variant": v, "tag": tag, "ms": ms, "tflops_equiv": tf})
        print(f"{name:12s} v{v:2d} {tag:24s} {ms:8.3f} ms {tf:8.1f} TF/s-equivalent", flush=True)


M = 66560
print("== start stagger, persistent kernel (variant 7)")
for phases, pct in [(0, 100), (2, 100), (4, 100), (8, 100), (4, 50), (2, 50), (3, 100), (0, 100)]:
    _lib.set_option("gemm_stagger_phases", phases)
    _lib.set_option("gemm_stagger_pct", pct)
    for shape in [("ffn1", M, 3072, 768, True), ("qk", M, 1536, 768, False), ("ffn2", M, 768, 3072, False)]:
        run(*shape, [7], tag=f"stagger {phases}x{pct}%")
_lib.set_option("gemm_stagger_phases", 0)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "gemm_stagger.json"), "w"), indent=1)
