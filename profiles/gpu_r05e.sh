#!/bin/bash
# Round 5: the encoder GEMM on v_mfma_f32_16x16x32_f16 (gemm_f16_p16.h) — the tests that run an encoder, per-shape A/B against the 32x32x16
# kernel, the new kernel's in-kernel ablations, A/B inside the encoder (BERT-base, NomicBert), micro-batch count, tail split.
# -> profiles/r05e_gemm_shapes_mfma16.json, r05e_gemm_p16_ablations.json, r05e_ab_encoder_mfma16.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_store_paths.py tests/test_gpu_nomic.py tests/test_gpu_rerank.py tests/test_gpu_splade.py tests/test_gpu_deberta.py tests/test_gpu_hf_path.py -x -q 2>&1 | tail -3 | tee gpurun_out/r05e_test.txt
timeout 500 python profiles/gemm_shapes_mfma16.py gpurun_out/r05e_gemm_shapes.json 0 1 2>&1 | grep -v amdgpu.ids
timeout 300 python profiles/gemm_p16_ablate.py gpurun_out/r05e_p16_ablate.json 2>&1 | grep -v amdgpu.ids
: > gpurun_out/r05e_ab.txt
timeout 300 python profiles/enc_ab_option.py gemm_mfma16 0 1 512 2>&1 | tail -3 | tee -a gpurun_out/r05e_ab.txt
ENC_ARCH=nomic timeout 300 python profiles/enc_ab_option.py gemm_mfma16 0 1 512 2>&1 | tail -3 | tee -a gpurun_out/r05e_ab.txt
timeout 300 python profiles/enc_ab_option.py micro_batches 1 2 512 2>&1 | tail -3 | tee -a gpurun_out/r05e_ab.txt
timeout 300 python profiles/enc_ab_option.py gemm_tail_split 0 1 512 2>&1 | tail -3 | tee -a gpurun_out/r05e_ab.txt
timeout 300 python profiles/stress_gemm_p16.py 40 2>&1 | grep -v amdgpu.ids | tail -3
