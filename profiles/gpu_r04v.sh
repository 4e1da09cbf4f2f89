#!/bin/bash
# attention: 16-byte context stores — parity of the attention kernel and the encoder, then the forward time of the bench batch
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
timeout 45 python -m pytest tests/test_gpu_encoder.py -m gpu -q -x --tb=short -p no:cacheprovider --timeout 40 -k "attention or golden_fixture or composition or micro_batches" 2>&1 | grep -v "amdgpu.ids" | tail -6 | cut -c1-300 | tee gpurun_out/r04v_pytest.txt
timeout 40 python profiles/enc_ab_option.py gemm_full_line_stores 0 1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04v_forward_ms.txt
