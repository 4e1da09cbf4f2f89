#!/bin/bash
# round 5: the 16x16x32 persistent GEMM as default + the transcendental-free GELU — every test that runs an encoder, the in-kernel ablations, forward timing
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_store_paths.py tests/test_gpu_nomic.py tests/test_gpu_rerank.py tests/test_gpu_splade.py tests/test_gpu_deberta.py tests/test_gpu_hf_path.py -x -q 2>&1 | tail -15 > gpurun_out/r05e_test.txt
tail -3 gpurun_out/r05e_test.txt
timeout 300 python profiles/gemm_p16_ablate.py gpurun_out/r05e_p16_ablate.json 2>&1 | grep -v amdgpu.ids | grep '"abl": 0,\|"abl": 8,\|"abl": 16,' | cut -c1-200
timeout 300 python profiles/enc_ab_option.py gemm_mfma16 0 1 512 2>&1 | tail -3 | tee gpurun_out/r05e_ab_bert.txt
