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
# the scan kernels of the run (bench.py times the 192-query kernel as the headline and the 128-query kernel as `tile128`);
# hbm_traffic.json describes the HEADLINE kernel (192-query tile when it ran) and names its query tile
scan = {k: cs for k, cs in res.items() if "bh_scan_topk" in k and "FETCH_SIZE" in cs}
for k, cs in scan.items():
    rd = cs["FETCH_SIZE"]["mean"] * 1024 * 2
    wr = cs.get("WRITE_SIZE", {"mean": 0})["mean"] * 1024
    tile = 192 if "topk192" in k else 128
    summary.setdefault("scan_hbm_bytes_per_launch", {})[str(tile)] = {
        "kernel": k, "read_corrected_x2": rd, "write": wr, "total": rd + wr, "fetch_size_raw": cs["FETCH_SIZE"]["mean"]}
if scan and len(sys.argv) > 4:
    head = max(scan, key=lambda k: ("topk192" in k, scan[k]["FETCH_SIZE"]["launches"]))
    h = summary["scan_hbm_bytes_per_launch"]["192" if "topk192" in head else "128"]
    json.dump({"n_rows": int(sys.argv[3]), "dim": int(sys.argv[4]), "query_tile": 192 if "topk192" in head else 128,
               "hbm_bytes_per_launch": h["total"], "kernel": head,
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), read side x2 per MI355X_MICROARCH.md"},
              open(os.path.join(os.path.dirname(out), "hbm_traffic.json"), "w"))
json.dump(summary, open(out, "w"), indent=1)
print(json.dumps(summary, indent=1)[:3000])
