#!/usr/bin/env python
"""profiles/<tag>_sq_summary.json (profiles/summarize_sq.py on a GPU box) -> profiles/sq_counters.json, the committed summary bench.py
reads `mfma_busy_frac` from (the way it reads `roofline.traffic` from profiles/hbm_traffic.json): per kernel instantiation the share of a
launch's shader-clock cycles in which a SIMD's matrix pipe was busy,

    mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)

(GRBM_GUI_ACTIVE arrives summed over the 8 XCDs: 177.8 M for a 13 ms launch at ~1.7 GHz = 8 x 22.2 M), the effective shader clock
implied by it when a launch duration is known, the wait fractions and the LDS conflict share.  Every entry carries the sha256 of the
kernel's source files at collection time: bench.py reports null for an entry whose source has changed since.
usage: python profiles/make_sq_counters.py profiles/r06a_sq_summary.json [more summaries ...]"""
import hashlib, json, os, sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "bergen_amd", "csrc")
SOURCES = {"bh_scan_topk256_kernel": ["scan_topk256.hip"], "bh_gemm_f16_p16kernel": ["gemm_f16_p16.h"], "bh_gemm_f16_pkernel": ["gemm_f16_persist.h"],
           "bh_attention_kernel": ["attention.hip"], "bh_layernorm_kernel": ["encoder_ops.hip"], "bh_csr_scan_mfma_kernel": ["csr_mfma.hip"]}


def sha16(files):
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


out = {"how": "rocprofv3 --pmc passes (SQ_* / GRBM_GUI_ACTIVE, kernel-filtered, no trace) -> profiles/summarize_sq.py -> this script",
       "formula": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * GRBM_GUI_ACTIVE / 8)", "kernels": {}}
for path in sys.argv[1:]:
    d = json.load(open(path))["per_kernel"]
    for k, r in d.items():
        c = r["counters"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c:
            continue
        base = next((b for b in SOURCES if b in k), None)
        if base is None:
            continue
        name = k.replace("void ", "").strip()
        cycles = c["GRBM_GUI_ACTIVE"]["mean"] / 8.0
        # over ALL launches of the instantiation (time-weighted) when the summary carries the sums; else the means of the full-size launches
        busy_frac = (c["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"] / (1024.0 * c["GRBM_GUI_ACTIVE"]["sum"] / 8.0)
                     if "sum" in c["GRBM_GUI_ACTIVE"] and c["GRBM_GUI_ACTIVE"]["sum"] else
                     c["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / (1024.0 * cycles) if cycles else None)
        e = {"mfma_busy_frac": busy_frac,
             "cycles_per_launch": cycles, "mfma_busy_cycles": c["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"],
             "launches": c["SQ_VALU_MFMA_BUSY_CYCLES"]["launches"], "source_files": SOURCES[base], "source_sha16": sha16(SOURCES[base]),
             "from": os.path.basename(path)}
        e.update({a: b for a, b in r["derived"].items() if a in ("wait_any_frac", "wait_inst_any_frac", "lds_conflict_frac")})
        if "SQ_BUSY_CYCLES" in c:
            e["sq_busy_cycles"] = c["SQ_BUSY_CYCLES"]["mean"]
        out["kernels"][name] = e
json.dump(out, open(os.path.join(HERE, "sq_counters.json"), "w"), indent=1)
for k, e in out["kernels"].items():
    print(f"{k[:70]:70s} mfma_busy_frac {e['mfma_busy_frac']:.3f}  cycles {e['cycles_per_launch']:.3g}")
