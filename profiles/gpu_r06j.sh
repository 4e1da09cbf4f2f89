#!/bin/bash
# knob sweep on the final encoder (BERT-base forward): tail split, side streams — with the 32-register LayerNorm in place
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
F="grep -v amdgpu.ids"
for rep in 1 2; do
  for kv in "micro_batches=2" "gemm_tail_split=1" "attn_side_stream=0" "vt_side_stream=0" "vt_side_stream=2"; do
    timeout 200 python profiles/enc_trace.py bert 10 $kv 2>&1 | $F | grep "forward ms" | cut -c1-110 | sed "s/^/$kv /"
  done
done | tee gpurun_out/r06j_encoder_knobs.txt
