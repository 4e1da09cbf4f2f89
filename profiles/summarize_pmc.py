#!/usr/bin/env python
"""Aggregate rocprofv3 counter_collection CSVs into a small per-kernel summary (run on the GPU box).
usage: summarize_pmc.py <prof_dir> <out.json> [n_rows dim]
HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB-like
units of 1024 B in rocprofv3's derived metric... we report the RAW counter and the corrected bytes:
  read_bytes  = FETCH_SIZE * 1024 * 2   (gfx950: FETCH_SIZE reports exactly 1/2 of a wide coalesced stream)
  write_bytes = WRITE_SIZE * 1024       (uncalibrated, small here)
"""
import csv, glob, json, os, sys
from collections import defaultdict

prof, out = sys.argv[1], sys.argv[2]
res = {}
for sub in sorted(os.listdir(prof)):
    for f in glob.glob(os.path.join(prof, sub, "*counter_collection.csv")):
        agg = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "?").split("(")[0][:60]
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in agg.items():
            for c, vals in cs.items():
                res.setdefault(k, {})[c] = {"launches": len(vals), "mean": sum(vals) / len(vals), "min": min(vals), "max": max(vals)}
summary = {"per_kernel": res}
# the scan kernels of the run (bench.py times the default kernel as the headline and the earlier ones as `other_kernels`);
# hbm_traffic.json is keyed by kernel name so that bench.py's roofline.traffic matches whatever kernel roofline.kernel names
def base_name(k):
    for name in ("bh_scan_topk256_kernel", "bh_scan_topk192_kernel", "bh_scan_topk_kernel"):
        if name in k:
            return name
    return None


scan = {k: cs for k, cs in res.items() if base_name(k) and "FETCH_SIZE" in cs}
traffic = {}
for k, cs in scan.items():
    rd = cs["FETCH_SIZE"]["mean"] * 1024 * 2
    wr = cs.get("WRITE_SIZE", {"mean": 0})["mean"] * 1024
    name = base_name(k)
    # a run may hold several instantiations of one kernel (ablations): keep the one with the most launches
    if name in traffic and traffic[name]["launches"] >= cs["FETCH_SIZE"]["launches"]:
        continue
    traffic[name] = {"kernel_full": k, "launches": cs["FETCH_SIZE"]["launches"], "read_corrected_x2": rd, "write": wr,
                     "hbm_bytes_per_launch": rd + wr, "fetch_size_raw": cs["FETCH_SIZE"]["mean"]}
summary["scan_hbm_bytes_per_launch"] = traffic
if traffic and len(sys.argv) > 4:
    for ent in traffic.values():
        ent["n_rows"] = int(sys.argv[3])
        ent["dim"] = int(sys.argv[4])
    json.dump({"kernels": traffic,
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), read side x2 per MI355X_MICROARCH.md"},
              open(os.path.join(os.path.dirname(out), "hbm_traffic.json"), "w"), indent=1)
json.dump(summary, open(out, "w"), indent=1)
print(json.dumps(summary, indent=1)[:3000])
