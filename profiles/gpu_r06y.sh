#!/bin/bash
# Round 6, last check of the final tree: the whole GPU suite, smoke, the bench with its default flags (what the driver runs)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
F="grep -v amdgpu.ids"
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 2>&1 | $F | tail -6 | cut -c1-300 | tee gpurun_out/r06_final_pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | $F | tail -1 | tee gpurun_out/r06_final_smoke.txt
( time timeout 600 python bench.py > gpurun_out/r06_final_bench.json 2> gpurun_out/r06_final_bench.err ) 2>&1 | grep real
python - <<'P'
import json
d = json.loads([l for l in open("gpurun_out/r06_final_bench.json") if l.startswith("{")][-1])
r = d["roofline"]
print(json.dumps({k: d.get(k) for k in ("value", "ms_per_step", "steps", "warmup", "parity_check")}))
print(json.dumps({k: r.get(k) for k in ("bound", "frac", "traffic", "frac_binding", "mfma_busy_frac", "avg_launch_ms")}))
print(json.dumps(d.get("encoder_roofline")))
P
