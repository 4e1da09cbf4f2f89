#!/bin/bash
# round 5, experiment: the 16x16x32 persistent GEMM (gemm_f16_p16.h) — parity, then A/B inside the encoder
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_encoder.py -x -q -k "mfma16" 2>&1 | tail -15 > gpurun_out/r05e_test.txt
cat gpurun_out/r05e_test.txt
for pair in "0 1" "1 2" "1 3" "1 4" "2 4"; do
  timeout 300 python profiles/enc_ab_option.py gemm_mfma16 $pair 512 2>&1 | tail -3 | tee -a gpurun_out/r05e_ab_bert.txt
done
