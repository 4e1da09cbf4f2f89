#!/bin/bash
# pair256: parity tests, then the same-box A/B (headline geometry and configs[4]; all rows and an eighth of them)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_search.py -m gpu -q -x -k "paired_launches" -p no:cacheprovider 2>&1 | tail -5
: > gpurun_out/r04j_ab_pair256.jsonl
for cfg in "1 1024 200" "8 1024 200" "1 768 50" "8 768 50"; do
  timeout 300 python profiles/ab_pair256.py $cfg 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r04j_ab_pair256.jsonl
done
