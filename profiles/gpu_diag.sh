#!/bin/bash
# GEMM ablations + rocprofv3 kernel trace of one encoder forward sweep (quick).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
timeout 300 python profiles/gemm_ablate.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gemm_ablate.log
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_enc" -o enc -- python $REPO/profiles/bench_encoder.py --quick > "$REPO/gpurun_out/prof_enc.log" 2>&1
echo "rocprof exit $?"
cd "$REPO"
find gpurun_out/prof_enc -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/enc_kernel_stats.csv
rm -rf gpurun_out/prof_enc
cut -c1-160 gpurun_out/enc_kernel_stats.csv | head -30
