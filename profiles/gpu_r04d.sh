#!/bin/bash
# Round 4: the 256-query filter pass of the exact fall-back — parity tests, then the certificate leg with the option on and off.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_search.py -q -x -p no:cacheprovider -k "fall_back or certificate or near_tie or large_k or repeatability or kat_small" 2>&1 | tail -6
B="python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-other-kernels --no-larger-k --no-config5 --no-real-size --no-stage --no-encoder --no-splade --encode-stage-passages 0 --full-list-queries 0 --no-power-leg"
timeout 600 $B > gpurun_out/r04d_bench_cert.json 2> gpurun_out/r04d_bench_cert.err; echo "bench exit $?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04d_bench_cert.json") if l.startswith("{")][-1])
print(json.dumps({"value": d["value"], "certificate": d.get("certificate")}, indent=1)[:3000])
PY
