#!/bin/bash
# round 5: SwiGLU fold on the 16x16x32 kernel — NomicBert / store-path tests, A/B; micro-batch count with the tail split
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_store_paths.py tests/test_gpu_nomic.py tests/test_gpu_encoder.py -x -q 2>&1 | tail -15 > gpurun_out/r05e_test.txt
tail -3 gpurun_out/r05e_test.txt
ENC_ARCH=nomic timeout 300 python profiles/enc_ab_option.py gemm_mfma16 0 1 512 2>&1 | tail -3 | tee gpurun_out/r05e_ab_nomic.txt
timeout 300 python profiles/enc_ab_option.py micro_batches 1 2 512 2>&1 | tail -3 | tee gpurun_out/r05e_ab_micro.txt
