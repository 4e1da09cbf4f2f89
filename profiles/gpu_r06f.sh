#!/bin/bash
# A/B of the 32-register LayerNorm kernel (option ln_small): encoder-only forward, interleaved, + its parity tests
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
F="grep -v amdgpu.ids"
timeout 600 python -m pytest tests/test_gpu_encoder.py -m gpu -q --tb=short -p no:cacheprovider --timeout 300 -k "layernorm or golden or batch" 2>&1 | $F | tail -5 | cut -c1-300
for rep in 1 2 3; do
  for v in 1 0; do
    timeout 200 python profiles/enc_trace.py bert 12 ln_small=$v 2>&1 | $F | grep "forward ms" | sed "s/^/ln_small=$v /"
  done
done | tee gpurun_out/r06f_ab_ln_small.txt
for v in 1 0; do timeout 200 python profiles/enc_trace.py e5_large 6 ln_small=$v 2>&1 | $F | grep "forward ms" | sed "s/^/ln_small=$v /"; done | tee -a gpurun_out/r06f_ab_ln_small.txt
for v in 1 0; do timeout 200 python profiles/enc_trace.py nomic 8 ln_small=$v 2>&1 | $F | grep "forward ms" | sed "s/^/ln_small=$v /"; done | tee -a gpurun_out/r06f_ab_ln_small.txt
