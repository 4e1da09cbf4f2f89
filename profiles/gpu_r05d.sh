#!/bin/bash
# Round 5: A/B of the encoder's existing scheduling knobs on the final store paths (same process, alternating), bit-identity included.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
F="grep -v amdgpu.ids"
: > gpurun_out/r05d_ab_encoder_knobs.txt
for spec in "vt_side_stream 1 2" "micro_batches 2 3" "gemm_gelu_nontemporal 1 0" "attn_side_stream 0 1"; do
  timeout 100 python profiles/enc_ab_option.py $spec 2>&1 | $F | tee -a gpurun_out/r05d_ab_encoder_knobs.txt
done
ENC_ARCH=nomic timeout 100 python profiles/enc_ab_option.py vt_side_stream 1 2 2>&1 | $F | tee -a gpurun_out/r05d_ab_encoder_knobs.txt
for v in 1 2; do timeout 120 python profiles/enc_trace.py e5_large 8 vt_side_stream=$v 2>&1 | $F | tail -1 | sed "s/^/vt_side_stream=$v /" | tee -a gpurun_out/r05d_ab_encoder_knobs.txt; done
