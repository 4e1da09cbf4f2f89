#!/bin/bash
# micro-batch count with the 32-register LayerNorm (option micro_batches 2 | 3 | 4), BERT-base encoder forward, interleaved
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
F="grep -v amdgpu.ids"
for rep in 1 2; do
  for mb in 2 3 4 1; do
    timeout 200 python profiles/enc_trace.py bert 10 micro_batches=$mb 2>&1 | $F | grep "forward ms" | sed "s/^/micro_batches=$mb /"
  done
done | tee gpurun_out/r06h_ab_micro_batches.txt
