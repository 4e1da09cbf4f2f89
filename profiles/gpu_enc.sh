#!/bin/bash
# Encoder-only GPU pass: encoder parity tests, GEMM configuration sweep + ablations, kernel sweep.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_encoder.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -n 30 | tee gpurun_out/pytest_enc.log
timeout 300 python profiles/gemm_ablate.py ${1:-} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gemm_ablate.log
timeout 600 python profiles/bench_encoder.py 2>&1 | grep -v amdgpu.ids | grep -E "forward|attention|Error|error" | tee gpurun_out/bench_encoder.log
