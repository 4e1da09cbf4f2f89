#!/bin/bash
# pair256: tiles between two checkpoints of partner workgroups — 8 / 16 (the build) / 32, same box, both geometries
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
: > gpurun_out/r04k_ab_pair_ckpt.jsonl
export AB_CONFIGS="0,1;1,1"
for cfg in "1 768 50" "1 1024 200"; do
  for lib in bergen_amd/lib profiles/bin/ckpt8 profiles/bin/ckpt32; do
    BERGEN_HIP_LIB=$REPO/$lib/libbergen_hip.so timeout 200 python profiles/ab_pair256.py $cfg 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04k_ab_pair_ckpt.jsonl
  done
done
