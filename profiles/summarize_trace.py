#!/usr/bin/env python
"""From a rocprofv3 kernel trace of `python bench.py ...`: every launch of the library's scan kernels in launch order
(scan_launches.csv: kernel, start, duration) and the average of the HEADLINE launches alone — the first (warmup + steps) x
(passes - 1) launches of the d = 768 instantiation of bh_scan_topk256_kernel, before any other leg of the bench reuses that
kernel (retrieve_stage_full, certificate) — which is the number bench.py's roofline.avg_launch_ms reports.
usage: summarize_trace.py <bench_kernel_trace.csv> <out_dir> <warmup + steps> <passes per step>"""
import csv
import json
import os
import sys

trace, out_dir, n_search, n_pass = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
rows = []
for r in csv.DictReader(open(trace)):
    name = r["Kernel_Name"]
    if "bh_scan_topk" in name or "bh_csr_scan" in name or "bh_merge_rescore" in name or "bh_exact" in name:
        rows.append((int(r["Start_Timestamp"]), name.split("(")[0].replace("void ", "")[:80], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows.sort()
t0 = rows[0][0] if rows else 0
with open(os.path.join(out_dir, "scan_launches.csv"), "w") as f:
    f.write("kernel,start_ms,duration_ms\n")
    for t, name, d in rows:
        f.write(f'"{name}",{(t - t0) / 1e6:.3f},{d / 1e6:.4f}\n')
def abl(name):
    """7th template argument of bh_scan_topk256_kernel<NK32, KP, LS, R, PD, NT, ABL, ...>: 128 = the paired launch (option pair256)."""
    parts = name.split("<", 1)[1].split(",")
    return int(parts[6]) if len(parts) > 6 and parts[6].strip().lstrip("-").isdigit() else 0


h64 = [(name, d) for _, name, d in rows if name.startswith("bh_scan_topk256_kernel<24, 64,")]  # (candidate lists of 64: k <= 56)
pair = [d for name, d in h64 if abl(name) == 128]
head = [d for name, d in h64 if abl(name) == 0]
tail = [d for _, name, d in rows if name.startswith("bh_scan_topk_kernel<48,") and ", 5, " not in name]
# a headline step of n_pass passes (the last one on the 128-query kernel): (n_pass - 1) // 2 paired launches + (n_pass - 1) % 2
# unpaired ones when option pair256 is on (any paired launch in the trace), n_pass - 1 unpaired ones otherwise
n_pair = n_search * ((n_pass - 1) // 2) if pair else 0
n_head = n_search * ((n_pass - 1) % 2 if pair else n_pass - 1)
res = {"headline_kernel": "bh_scan_topk256_kernel<24, 64, ...> (d = 768)",
       "paired_launches": n_pair, "paired_avg_ms": sum(pair[:n_pair]) / max(1, len(pair[:n_pair])) / 1e6,
       "all_paired_launches_of_that_instantiation": len(pair), "all_paired_avg_ms": sum(pair) / max(1, len(pair)) / 1e6,
       "headline_launches": n_head,
       "headline_avg_ms": sum(head[:n_head]) / max(1, len(head[:n_head])) / 1e6,
       "all_launches_of_that_instantiation": len(head), "all_avg_ms": sum(head) / max(1, len(head)) / 1e6,
       "tail_pass_kernel": "bh_scan_topk_kernel<48, ...> (128-query kernel, last pass of a step)",
       "tail_pass_avg_ms_first_searches": sum(tail[:n_search]) / max(1, len(tail[:n_search])) / 1e6}
json.dump(res, open(os.path.join(out_dir, "headline_from_trace.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
