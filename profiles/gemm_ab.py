#!/usr/bin/env python
"""A/B of encoder GEMM variants on the five projection shapes of a BERT-base layer (M = 68 608 packed rows of the bench
batch), interleaved rounds in one process:  python profiles/gemm_ab.py [out.json] [variants ...]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bergen_amd import encoder  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "gemm_ab.json")
    variants = [int(v) for v in sys.argv[2:]] or [7, 33, 0]
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)
    M = 68608
    shapes = [("qk", M, 1536, 768, False, 1), ("vt", 768, M, 768, False, 2), ("attn_out", M, 768, 768, False, 1),
              ("ffn1", M, 3072, 768, True, 1), ("ffn2", M, 768, 3072, False, 1)]
    res = []
    for name, m, n, k, gelu, bias_mode in shapes:
        a = (torch.randn(m, k, generator=g, device=dev) * 0.5).half()
        w = (torch.randn(n, k, generator=g, device=dev) * 0.05).half()
        bias = (torch.randn(m if bias_mode == 2 else n, generator=g, device=dev) * 0.1).half()
        outb = torch.empty(m, n, dtype=torch.float16, device=dev)
        times = {v: [] for v in variants}
        for rnd in range(5):
            for v in variants:
                _, ms = encoder.gemm_f16(a, w, bias=bias, bias_mode=bias_mode, gelu=gelu, variant=v, out=outb, repeats=7)
                times[v].append(ms)
        row = {"shape": name, "M": m, "N": n, "K": k}
        for v in variants:
            t = sorted(times[v])
            row[f"v{v}_ms_median"] = t[len(t) // 2]
            row[f"v{v}_ms_min"] = t[0]
            row[f"v{v}_tflops"] = 2.0 * m * n * k / (t[len(t) // 2] * 1e-3) / 1e12
        res.append(row)
        print(json.dumps(row), flush=True)
        del a, w, outb
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
