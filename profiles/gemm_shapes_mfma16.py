#!/usr/bin/env python
"""Per-shape A/B of option gemm_mfma16 (gemm_f16_p16.h) on the row-major projections of a BERT-base / bert-large layer, at the
bench batch's micro-batch row count and at row counts that fill whole rounds of 256 tiles:
  python profiles/gemm_shapes_mfma16.py [out.json] [values ...]"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bergen_amd import _lib, encoder  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "gemm_shapes_mfma16.json")
values = [int(v) for v in sys.argv[2:]] or [0, 2]
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
shapes = []
for M in (33280, 32768, 66560, 65536):
    shapes += [("qk", M, 1536, 768, False), ("attn_out", M, 768, 768, False), ("ffn1", M, 3072, 768, True), ("ffn2", M, 768, 3072, False)]
shapes += [("L_qk", 33280, 2048, 1024, False), ("L_out", 33280, 1024, 1024, False), ("L_ffn1", 33280, 4096, 1024, True), ("L_ffn2", 33280, 1024, 4096, False)]
res = []
for name, m, n, k, gelu in shapes:
    a = (torch.randn(m, k, generator=g, device=dev) * 0.5).half()
    w = (torch.randn(n, k, generator=g, device=dev) * 0.05).half()
    bias = (torch.randn(n, generator=g, device=dev) * 0.1).half()
    outb = torch.empty(m, n, dtype=torch.float16, device=dev)
    times = {v: [] for v in values}
    for rnd in range(5):
        for v in values:
            _lib.set_option("gemm_mfma16", v)
            _, ms = encoder.gemm_f16(a, w, bias=bias, bias_mode=1, gelu=gelu, out=outb, repeats=7)
            times[v].append(ms)
    tiles = (m // 256) * (n // 256)
    row = {"shape": name, "M": m, "N": n, "K": k, "tiles": tiles, "rounds": round(tiles / 256, 2)}
    for v in values:
        t = sorted(times[v])
        row[f"mfma16={v}_ms"] = round(t[len(t) // 2], 4)
        row[f"mfma16={v}_tflops"] = round(2.0 * m * n * k / (t[len(t) // 2] * 1e-3) / 1e12, 1)
    res.append(row)
    print(json.dumps(row), flush=True)
    del a, w, outb
_lib.set_option("gemm_mfma16", 1)  # (the library default)
json.dump(res, open(out, "w"), indent=1)
