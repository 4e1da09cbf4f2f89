#!/usr/bin/env python
"""Does the row stride of the token operand matter to the GEMM's refill?  The FFN-down shape reads rows 6 144 bytes apart (24 x 256 B): if the
L2 channel of an address were (address / 256) mod 16, the 256 row segments of a stage would sit on TWO channels.  Same GEMM with the A rows padded
by 0 / 64 / 128 / 192 / 256 elements; production kernel and its LDS-DMA-only ablation (GELU epilogue instantiation, results ignored).
  python profiles/gemm_row_stride.py [out.json]"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bergen_amd import _lib, encoder  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "gemm_row_stride.json")
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
res = []
for (m, n, k) in [(65536, 768, 3072), (65536, 3072, 768), (65536, 1536, 768), (33280, 1024, 4096)]:
    w = (torch.randn(n, k, generator=g, device=dev) * 0.05).half()
    bias = (torch.randn(n, generator=g, device=dev) * 0.1).half()
    outb = torch.empty(m, n, dtype=torch.float16, device=dev)
    pads = (0, 64, 128, 192, 256)
    bufs = {p: (torch.randn(m, k + p, generator=g, device=dev) * 0.5).half() for p in pads}
    times = {}
    for rnd in range(3):
        for p in pads:
            a = bufs[p][:, :k]
            for mode, gelu in ((1, False), (16 * 14 + 1, True), (16 * 8 + 1, True)):
                _lib.set_option("gemm_mfma16", mode)
                _, ms = encoder.gemm_f16(a, w, bias=bias, bias_mode=1, gelu=gelu, out=outb, repeats=7)
                times.setdefault((p, mode), []).append(ms)
    for p in pads:
        row = {"M": m, "N": n, "K": k, "a_row_pad_elements": p, "a_row_stride_bytes": 2 * (k + p)}
        for mode, name in ((1, "production_ms"), (16 * 14 + 1, "lds_dma_only_ms"), (16 * 8 + 1, "no_epilogue_ms")):
            t = sorted(times[(p, mode)])
            row[name] = round(t[len(t) // 2], 4)
        res.append(row)
        print(json.dumps(row), flush=True)
    del bufs, w, outb
_lib.set_option("gemm_mfma16", 1)
json.dump(res, open(out, "w"), indent=1)
