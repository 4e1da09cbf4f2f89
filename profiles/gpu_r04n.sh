#!/bin/bash
# host-side gap of a headline step on a (possibly loaded) box: glibc's default malloc thresholds vs no mmap / no trim for the result copies
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
echo "== default malloc" | tee gpurun_out/r04n_step_gap.txt
timeout 150 python profiles/step_gap.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04n_step_gap.txt
echo "== MALLOC_MMAP_THRESHOLD_=32M MALLOC_TRIM_THRESHOLD_=512M MALLOC_TOP_PAD_=64M" | tee -a gpurun_out/r04n_step_gap.txt
MALLOC_MMAP_THRESHOLD_=33554432 MALLOC_TRIM_THRESHOLD_=536870912 MALLOC_TOP_PAD_=67108864 timeout 150 python profiles/step_gap.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04n_step_gap.txt
