#!/bin/bash
# Kernel-time budget of the rerank leg (two cross-encoders, 32 pairs): sum of kernel durations vs the forward time.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/rerank_trace
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o rr -- python $REPO/profiles/rerank_only.py > $OUT/log.txt 2>&1
cp $OUT/t/rr_kernel_stats.csv $OUT/kernel_stats.csv
python - <<PY
import csv, collections
rows=list(csv.DictReader(open("$OUT/t/rr_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
starts=[i for i,r in enumerate(rows) if r["Kernel_Name"].startswith("bh_embed_ln")]
print("launches",len(rows),"forwards",len(starts))
segs=[(starts[i], starts[i+1] if i+1<len(starts) else len(rows)) for i in range(len(starts))]
def is_rel(seg): return any("attention_rel" in r["Kernel_Name"] for r in rows[seg[0]:seg[1]])
for label,want in (("deberta",True),("bert_large",False)):
    cand=[s for s in segs if is_rel(s)==want]
    if not cand: continue
    a,b=cand[-1]
    seg=rows[a:b]
    # cut at the classification head (what follows belongs to the host side)
    for i,r in enumerate(seg):
        if "cls_head" in r["Kernel_Name"]: seg=seg[:i+1]; break
    busy=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in seg)
    span=int(seg[-1]["End_Timestamp"])-int(seg[0]["Start_Timestamp"])
    print("== %s last forward: %d launches, busy %.3f ms, span %.3f ms"%(label,len(seg),busy/1e6,span/1e6))
    by=collections.Counter(); cnt=collections.Counter()
    for r in seg:
        k=r["Kernel_Name"][:58]; by[k]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"]); cnt[k]+=1
    for k,v in by.most_common(12): print("  %-60s %4d %7.3f ms avg %6.1f us"%(k,cnt[k],v/1e6,v/cnt[k]/1e3))
PY
rm -rf $OUT/t
