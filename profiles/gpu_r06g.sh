#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_r06g
mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
for v in 1 0; do
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT/t$v" -o enc -- python "$REPO/profiles/enc_trace.py" bert 4 ln_small=$v > "$OUT/log$v.txt" 2>&1
  f=$(find "$OUT/t$v" -name "*kernel_trace.csv" | head -1)
  echo "ln_small=$v: $(grep 'forward ms' $OUT/log$v.txt)"
  python "$REPO/profiles/ln_overlap.py" "$f"
  rm -rf "$OUT/t$v"
done
