#!/bin/bash
# round 5: the SPLADE head's segmented-max epilogue on the 16x16x32 kernel — SPLADE tests, head cost with gemm_mfma16 0 / 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_splade.py tests/test_gpu_encoder.py -x -q 2>&1 | tail -4 | tee gpurun_out/r05h_test.txt
: > gpurun_out/r05h_splade_head.jsonl
for v in 0 1 0 1; do
  BERGEN_GEMM_MFMA16=$v timeout 200 python - <<P 2>&1 | grep '^{' | sed "s/^/{\"gemm_mfma16\": $v, \"r\": /; s/$/}/" | tee -a gpurun_out/r05h_splade_head.jsonl
import os, sys, runpy
sys.path.insert(0, ".")
from bergen_amd import _lib
_lib.set_option("gemm_mfma16", int(os.environ["BERGEN_GEMM_MFMA16"]))
sys.argv = ["bench_splade_encode.py", "512", "8"]
runpy.run_path("profiles/bench_splade_encode.py", run_name="__main__")
P
done
