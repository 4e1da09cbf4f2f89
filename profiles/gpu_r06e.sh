#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_r06e
mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t" -o rr -- python "$REPO/profiles/deberta_trace.py" 256 > "$OUT/log.txt" 2>&1
grep "pairs 256" "$OUT/log.txt"
cp "$OUT/t/rr_kernel_stats.csv" "$REPO/gpurun_out/r06_deberta256_kernel_stats.csv"
head -14 "$REPO/gpurun_out/r06_deberta256_kernel_stats.csv" | cut -c1-150
rm -rf "$OUT/t"
cd "$REPO"
python profiles/merge_bench.py 2>&1 | grep -v amdgpu.ids | tail -12
