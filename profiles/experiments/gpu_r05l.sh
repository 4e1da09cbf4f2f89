#!/bin/bash
# round 5, experiment: gemm_f16_p16a.h (token rows one stage further ahead) — parity / race screen, per-shape and encoder A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python profiles/check_gemm_p16a.py 10 2>&1 | grep -v amdgpu | tail -12 | tee gpurun_out/r05l_check.txt
timeout 300 python profiles/gemm_shapes_mfma16.py gpurun_out/r05l_gemm_shapes.json 1 5 2>&1 | grep -v amdgpu.ids | cut -c1-260
timeout 300 python profiles/enc_ab_option.py gemm_mfma16 1 5 512 2>&1 | tail -3 | tee gpurun_out/r05l_ab_bert.txt
