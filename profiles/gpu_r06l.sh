#!/bin/bash
# stream priority of the second micro-batch's stream (BERGEN_AMD_MB_PRIORITY) x the 32-register LayerNorm: does anti-phase scheduling hide LayerNorm under the other stream's GEMM?
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
F="grep -v amdgpu.ids"
for rep in 1 2; do
  for pr in 0 1 -1; do
    for ln in 1 0; do
      BERGEN_AMD_MB_PRIORITY=$pr timeout 200 python profiles/enc_trace.py bert 10 ln_small=$ln 2>&1 | $F | grep "forward ms" | cut -c1-110 | sed "s/^/mb_priority=$pr ln_small=$ln /"
    done
  done
done | tee gpurun_out/r06l_ab_mb_priority.txt
