#!/bin/bash
# rocprofv3 kernel trace of the encoder forward (bench_encoder.py --quick): per-kernel time breakdown.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$REPO/gpurun_out"
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_enc" -o enc -- python $REPO/profiles/bench_encoder.py --quick --no-gemm-sweep > "$REPO/gpurun_out/prof_enc.log" 2>&1
echo "rocprof exit $?"
cd "$REPO"
find gpurun_out/prof_enc -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/enc_kernel_stats.csv
rm -rf gpurun_out/prof_enc
cut -c1-150 gpurun_out/enc_kernel_stats.csv | head -16
