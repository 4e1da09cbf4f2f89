#!/bin/bash
# First GPU call of round 5: the store paths finished in round 4's last GPU seconds (tests/test_gpu_experimental.py: green once), their A/B
# on the encoder legs, then — with the options ON through the environment of a full `pytest -m gpu` run — the step that makes them defaults.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_gpu_experimental.py -m gpu -q --tb=short -p no:cacheprovider --timeout 150 2>&1 | grep -v amdgpu.ids | tail -15 | cut -c1-300 | tee gpurun_out/r05_experimental_pytest.txt
timeout 100 python profiles/enc_ab_option.py gemm_full_line_stores 1 2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_ab_full_line_level2.txt
ENC_ARCH=nomic timeout 100 python profiles/enc_ab_option.py gemm_full_line_stores 1 2 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r05_ab_full_line_level2.txt
# A/B knobs that change no bits: where the FFN-up output goes (caches or not) now that a micro-batch writes 210 MB, and the micro-batch count under the new store paths
timeout 100 python profiles/enc_ab_option.py gemm_gelu_nontemporal 1 0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_ab_knobs.txt
timeout 100 python profiles/enc_ab_option.py micro_batches 2 3 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r05_ab_knobs.txt
