"""Timeline of the LAST search in a rocprofv3 kernel trace CSV: start / end of every kernel relative to the first, in us."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last search starts at the last bh_fill_u32 launch
last_fill = max(i for i, r in enumerate(rows) if "bh_fill_u32" in r["Kernel_Name"])
# back up over the query conversion in front of it
start = last_fill
while start > 0 and ("convert" in rows[start - 1]["Kernel_Name"] or "normalize" in rows[start - 1]["Kernel_Name"]):
    start -= 1
t0 = int(rows[start]["Start_Timestamp"])
prev_end = t0
for r in rows[start:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f}  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  {name}")
    prev_end = max(prev_end, e)
