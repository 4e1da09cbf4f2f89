#!/bin/bash
# encoder GEMM: outputs as whole 128-byte lines through LDS (option gemm_full_line_stores) — bit-identity, then A/B on both encoder legs
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
timeout 120 python -m pytest "tests/test_gpu_encoder.py::test_gemm_full_line_stores_are_bit_identical" "tests/test_gpu_encoder.py::test_gemm_persistent_many_tiles_per_block" -m gpu -q --tb=short -p no:cacheprovider --timeout 100 2>&1 | grep -v amdgpu.ids | tail -8 | cut -c1-300 | tee gpurun_out/r04t_pytest.txt
timeout 100 python profiles/enc_ab_option.py gemm_full_line_stores 0 1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04t_ab_full_line_stores.txt
ENC_ARCH=nomic timeout 100 python profiles/enc_ab_option.py gemm_full_line_stores 0 1 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04t_ab_full_line_stores.txt
