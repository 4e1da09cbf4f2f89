#!/bin/bash
# NomicBert on the HIP path: kernel-level and encoder-level parity (rotary positions, gated SiLU feed-forward), plus the BERT-path
# tests next to the code that changed (encoder.hip orchestration, ABI structs)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
timeout 280 python -m pytest tests/test_gpu_nomic.py tests/test_gpu_abi.py "tests/test_gpu_encoder.py::test_encoder_matches_hf_golden_fixture" "tests/test_gpu_encoder.py::test_micro_batches_give_identical_outputs" "tests/test_gpu_encoder.py::test_encoder_bert_base_shape_against_oracle" tests/test_gpu_splade.py -m gpu -q --tb=short -p no:cacheprovider -s --timeout 120 2>&1 | grep -v amdgpu.ids | tail -40 | cut -c1-300 | tee gpurun_out/r04p_pytest_nomic.txt
