#!/bin/bash
# Round 5, second GPU call: the LayerNorm passes fused into the GEMM epilogues (encoder option ln_fused) — parity, then the same-process A/B.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
export TMPDIR=/tmp
F="grep -v amdgpu.ids"
timeout 500 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_store_paths.py tests/test_gpu_splade.py tests/test_gpu_rerank.py tests/test_gpu_hf_path.py tests/test_gpu_ut1.py \
    -m gpu -q -x --tb=short -p no:cacheprovider --timeout 300 --durations=5 2>&1 | $F | tail -30 | cut -c1-500 | tee gpurun_out/r05b_pytest_encoder.txt
timeout 120 python profiles/enc_ab_option.py ln_fused 0 1 2>&1 | $F | tee gpurun_out/r05b_ab_ln_fused.txt
for v in 0 1; do timeout 120 python profiles/enc_trace.py e5_large 8 ln_fused=$v 2>&1 | $F | tail -1 | sed "s/^/ln_fused=$v /" | tee -a gpurun_out/r05b_ab_ln_fused.txt; done
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_r05b_enc" -o enc -- python "$REPO/profiles/enc_trace.py" bert 10 > "$REPO/gpurun_out/r05b_enc_trace.log" 2>&1; echo "rocprof exit $?")
find gpurun_out/prof_r05b_enc -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r05b_encoder_kernel_stats.csv
rm -rf gpurun_out/prof_r05b_enc
$F gpurun_out/r05b_enc_trace.log | tail -2
cut -c1-160 gpurun_out/r05b_encoder_kernel_stats.csv | head -16
