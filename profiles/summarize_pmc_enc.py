#!/usr/bin/env python
"""Per-kernel means of the SQ counters rocprofv3 --pmc collected for the encoder kernels (GEMM, attention)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

src, out = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        short = name.split("(")[0][:90]
        acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {}
for k, ctrs in acc.items():
    res[k] = {c: {"mean": sum(v) / len(v), "launches": len(v)} for c, v in ctrs.items()}
    m = res[k]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CYCLES" in m and m["SQ_BUSY_CYCLES"]["mean"] > 0:
        # MFMA busy cycles summed over SIMDs / (busy cycles of the SQs x 4 SIMDs per CU ... reported raw: see README)
        m["_mfma_busy_over_sq_busy"] = m["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / m["SQ_BUSY_CYCLES"]["mean"]
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: {c: round(v["mean"]) if isinstance(v, dict) else v for c, v in m.items()} for k, m in res.items()}, indent=1)[:3000])
