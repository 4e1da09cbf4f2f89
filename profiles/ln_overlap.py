"""From a rocprofv3 kernel trace of profiles/enc_trace.py: how much of the LayerNorm kernels' time overlaps a GEMM kernel of the other stream.
usage: python profiles/ln_overlap.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
gemm = sorted((s, e) for s, e, n in ev if "bh_gemm" in n)
ln = [(s, e, n) for s, e, n in ev if "layernorm" in n]
tot = ov = 0
durs = []
for s, e, n in ln:
    tot += e - s
    durs.append(e - s)
    for gs, ge in gemm:
        if ge <= s: continue
        if gs >= e: break
        ov += min(e, ge) - max(s, gs)
durs.sort()
names = sorted({n for _, _, n in ln})
print(f"LayerNorm launches {len(ln)} ({names}), total {tot / 1e6:.3f} ms, overlapped with a GEMM launch {ov / 1e6:.3f} ms ({100.0 * ov / max(1, tot):.1f} %), "
      f"duration median {durs[len(durs) // 2] / 1e3:.1f} us, p10 {durs[len(durs) // 10] / 1e3:.1f}, p90 {durs[9 * len(durs) // 10] / 1e3:.1f}")
span = max(e for _, e, _ in ev) - min(s for s, _, _ in ev)
busy = sum(e - s for s, e, _ in ev)
print(f"all kernels: span {span / 1e6:.3f} ms, sum of durations {busy / 1e6:.3f} ms")
