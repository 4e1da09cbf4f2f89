#!/bin/bash
# round 5: L2 prefetch of the token rows in gemm_f16_p16.h (gemm_mfma16 = 3) — parity, ablations, per-shape and encoder A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_encoder.py -x -q -k "mfma16 or tail_split" 2>&1 | tail -3 | tee gpurun_out/r05i_test.txt
timeout 300 python profiles/gemm_p16_ablate.py gpurun_out/r05i_p16_ablate.json 2>&1 | grep -v amdgpu.ids | grep '"abl": 0,\|"abl": 8,\|"abl": 14,' | cut -c1-220
timeout 300 python profiles/gemm_shapes_mfma16.py gpurun_out/r05i_gemm_shapes.json 1 3 2>&1 | grep -v amdgpu.ids | cut -c1-260
timeout 300 python profiles/enc_ab_option.py gemm_mfma16 1 3 512 2>&1 | tail -3 | tee gpurun_out/r05i_ab_bert.txt
