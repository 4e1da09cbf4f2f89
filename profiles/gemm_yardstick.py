#!/usr/bin/env python
"""External yardstick for the encoder GEMMs: the VENDOR library (hipBLASLt / rocBLAS behind
torch.mm / torch.nn.functional.linear) on the five projection shapes of a BERT-base layer at the bench batch (M = 68 608
packed rows), same box, same process, interleaved rounds with this repository's persistent kernel (variant 7).

    python profiles/gemm_yardstick.py [out.json]

PROFILES ONLY.  torch's GEMM is never called from bergen_amd/ (tests/test_abi.py greps for it); this script exists to tell
whether the "three ceilings" reading of DESIGN.md (main loop balanced on the matrix pipe, the LDS-DMA path and the store path)
is a property of the shapes or of our kernel.  Reported per shape:
  ours_ms            bh_op_gemm_f16 with its fused epilogue (bias, GELU for FFN-up)
  vendor_mm_ms       torch.mm(a, w.T) alone                       (the vendor's main loop + plain store)
  vendor_fused_ms    F.linear(a, w, bias) (+ F.gelu as a 2nd op)  (what an unfused torch forward pays)
random operands (never zeros: DVFS, guide rule 25)."""
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bergen_amd import encoder  # noqa: E402


def timed(fn, reps=7):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "gemm_yardstick.json")
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)
    M = int(os.environ.get("BH_M", "68608"))
    # name, m, n, k, gelu, bias_mode (1 = per column, 2 = per row: the V projection is computed transposed, weights as A)
    shapes = [("qk", M, 1536, 768, False, 1), ("vt", 768, M, 768, False, 2), ("attn_out", M, 768, 768, False, 1),
              ("ffn1", M, 3072, 768, True, 1), ("ffn2", M, 768, 3072, False, 1),
              ("qkv_fused", M, 2304, 768, False, 1),          # one N = 2304 launch instead of qk + vt
              ("square_4096", 4096, 4096, 4096, False, 1)]    # the guide's reference shape, for scale
    res = []
    for name, m, n, k, gelu, bias_mode in shapes:
        a = (torch.randn(m, k, generator=g, device=dev) * 0.5).half()
        w = (torch.randn(n, k, generator=g, device=dev) * 0.05).half()
        bias = (torch.randn(m if bias_mode == 2 else n, generator=g, device=dev) * 0.1).half()
        outb = torch.empty(m, n, dtype=torch.float16, device=dev)
        wt = w.t()
        t = {"ours": [], "mm": [], "fused": []}
        for rnd in range(5):
            _, ms = encoder.gemm_f16(a, w, bias=bias, bias_mode=bias_mode, gelu=gelu, variant=7, out=outb, repeats=7)
            t["ours"].append(ms)
            t["mm"].append(timed(lambda: torch.mm(a, wt, out=outb)))
            if bias_mode == 1:
                t["fused"].append(timed((lambda: F.gelu(F.linear(a, w, bias))) if gelu else (lambda: F.linear(a, w, bias))))
            else:
                t["fused"].append(timed(lambda: torch.addmm(bias[:, None], a, wt)))
        flop = 2.0 * m * n * k
        row = {"shape": name, "M": m, "N": n, "K": k, "gelu": gelu}
        for key, label in (("ours", "ours"), ("mm", "vendor_mm"), ("fused", "vendor_fused")):
            v = sorted(t[key])
            row[f"{label}_ms"] = v[len(v) // 2]
            row[f"{label}_ms_min"] = v[0]
            row[f"{label}_tflops"] = flop / (v[len(v) // 2] * 1e-3) / 1e12
        row["vendor_mm_over_ours"] = row["ours_ms"] / row["vendor_mm_ms"]
        res.append(row)
        print(json.dumps(row), flush=True)
        del a, w, outb, wt
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump({"torch": torch.__version__, "hip": torch.version.hip, "rows": res}, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
