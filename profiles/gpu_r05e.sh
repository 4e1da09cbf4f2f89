#!/bin/bash
# round 5, experiment: the 16x16x32 persistent GEMM (gemm_f16_p16.h) as the default — encoder / store-path / rerank / splade-encode tests, A/B inside the encoder
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_store_paths.py tests/test_gpu_nomic.py tests/test_gpu_rerank.py tests/test_gpu_splade.py tests/test_gpu_deberta.py tests/test_gpu_hf_path.py -x -q 2>&1 | tail -15 > gpurun_out/r05e_test.txt
cat gpurun_out/r05e_test.txt
: > gpurun_out/r05e_ab_bert.txt
timeout 300 python profiles/enc_ab_option.py gemm_mfma16 0 1 512 2>&1 | tail -3 | tee -a gpurun_out/r05e_ab_bert.txt
timeout 300 python profiles/enc_ab_option.py gemm_tail_split 0 1 512 2>&1 | tail -3 | tee -a gpurun_out/r05e_ab_bert.txt
ENC_ARCH=nomic timeout 300 python profiles/enc_ab_option.py gemm_mfma16 0 1 512 2>&1 | tail -3 | tee -a gpurun_out/r05e_ab_bert.txt
