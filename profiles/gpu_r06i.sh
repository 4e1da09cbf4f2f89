#!/bin/bash
# rotary positions in the Q | K GEMM's epilogue: parity tests + same-box A/B on the NomicBert and gte encoder forward
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
F="grep -v amdgpu.ids"
timeout 900 python -m pytest tests/test_gpu_nomic.py tests/test_gpu_gte.py tests/test_gpu_encoder.py tests/test_gpu_store_paths.py -m gpu -q --tb=short -p no:cacheprovider --timeout 300 2>&1 | $F | grep -E "^FAILED|^E  |passed|failed" | head -12 | cut -c1-300
for rep in 1 2; do
  for arch in nomic gte; do
    for v in 1 0; do
      timeout 200 python profiles/enc_trace.py $arch 10 gemm_rotary_fused=$v 2>&1 | $F | grep "forward ms" | cut -c1-130 | sed "s/^/rotary_fused=$v /"
    done
  done
done | tee gpurun_out/r06i_ab_rotary_fused.txt
