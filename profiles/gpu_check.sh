#!/bin/bash
# One GPU-box pass: parity tests, encoder kernel sweep, headline bench.  Logs land in gpurun_out/.
#   gpurun --timeout 1500 -- 'bash profiles/gpu_check.sh'
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
mkdir -p gpurun_out
echo "== pytest -m gpu"; date
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a gpurun_out/pytest_gpu.log
tail -n 40 gpurun_out/pytest_gpu.log
echo "== encoder kernel sweep"; date
timeout 600 python profiles/bench_encoder.py > gpurun_out/bench_encoder.log 2>&1
echo "bench_encoder exit $?" | tee -a gpurun_out/bench_encoder.log
tail -n 60 gpurun_out/bench_encoder.log
echo "== bench.py"; date
timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 > gpurun_out/bench.log 2> gpurun_out/bench.err
echo "bench exit $?" | tee -a gpurun_out/bench.err
cat gpurun_out/bench.log; tail -n 5 gpurun_out/bench.err
date
