#!/usr/bin/env python
"""SQ / GRBM counters of rocprofv3 --pmc passes -> per-kernel means and the derived fractions bench.py reports beside its rooflines
(`mfma_busy_frac`).  usage: summarize_sq.py <prof_dir with one sub-directory per pass> <out.json>

Derived per kernel (means over the full-size launches of the kernel = those whose SQ_WAVE_CYCLES / GRBM_GUI_ACTIVE is >= half the largest):
  cycles          = GRBM_GUI_ACTIVE (shader-clock cycles the launch was resident; summed over XCDs by rocprofv3 when it says so: the
                    script divides by 8 when the value exceeds 4x what the kernel's wall time allows at 2.4 GHz — recorded in `gui_div`)
  mfma_busy_frac  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles): the share of the launch's cycles a SIMD's matrix pipe was busy
  mfma_busy_frac_of_sq_busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / SE_COUNT x 1024 / ...) is NOT computed: SQ_BUSY_CYCLES' aggregation
                    over shader engines differs between rocprofv3 versions; the raw value is kept.
  wait_any_frac   = SQ_WAIT_ANY / SQ_WAVE_CYCLES, wait_inst_any_frac = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (both quad-cycle counters)
  lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
"""
import csv, glob, json, os, sys
from collections import defaultdict

prof, out = sys.argv[1], sys.argv[2]
vals = defaultdict(lambda: defaultdict(list))
for sub in sorted(os.listdir(prof)):
    for f in glob.glob(os.path.join(prof, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "?").split("(")[0][:90]
            vals[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {}
for k, cs in vals.items():
    e = {}
    for c, v in cs.items():
        big = [x for x in v if x >= 0.5 * max(v)] or v
        e[c] = {"launches": len(big), "mean": sum(big) / len(big), "min": min(v), "max": max(v), "launches_all": len(v), "sum": sum(v)}
    d = {}
    m = lambda c: e[c]["mean"] if c in e else None
    if m("SQ_VALU_MFMA_BUSY_CYCLES") is not None and m("GRBM_GUI_ACTIVE"):
        gui = m("GRBM_GUI_ACTIVE")
        d["gui_active_cycles_raw"] = gui
        d["mfma_busy_cycles_per_simd"] = m("SQ_VALU_MFMA_BUSY_CYCLES") / 1024
    if m("SQ_WAVE_CYCLES"):
        for c, name in (("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_WAIT_INST_ANY", "wait_inst_any_frac")):
            if m(c) is not None:
                d[name] = m(c) / m("SQ_WAVE_CYCLES")
    if m("SQ_LDS_IDX_ACTIVE"):
        d["lds_conflict_frac"] = (m("SQ_LDS_BANK_CONFLICT") or 0.0) / m("SQ_LDS_IDX_ACTIVE")
    if m("SQ_INSTS_VALU_MFMA_MOPS_F16") is not None:
        d["mfma_mops_f16"] = m("SQ_INSTS_VALU_MFMA_MOPS_F16")
    res[k] = {"counters": e, "derived": d}
json.dump({"per_kernel": res, "how": "profiles/gpu_r06a.sh; rocprofv3 --pmc, two passes of 8 counters, kernel-filtered, no trace"}, open(out, "w"), indent=1)
for k, r in sorted(res.items()):
    print(k[:80], {a: (round(b, 4) if b < 10 else round(b)) for a, b in r["derived"].items()},
          {c: round(r["counters"][c]["mean"]) for c in ("SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE") if c in r["counters"]},
          "launches", max(x["launches"] for x in r["counters"].values()))
