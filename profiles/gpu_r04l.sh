#!/bin/bash
# Round 4, last pass, call 1 of 2: every GPU test on the build with paired launches on by default (ABI 141), then the bench's search legs
# (headline, configs[4] geometry, real row count: the three that report through bench.scan_roofline) as a quick check of the line.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
mkdir -p gpurun_out
timeout 560 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=6 --timeout 200 > gpurun_out/r04l_pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a gpurun_out/r04l_pytest_gpu.log
tail -n 30 gpurun_out/r04l_pytest_gpu.log | cut -c1-400
timeout 240 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-encoder --no-splade --no-stage --no-certificate-leg --no-larger-k --no-other-kernels --encode-stage-passages 0 > gpurun_out/r04l_bench_search_legs.json 2> gpurun_out/r04l_bench.err
echo "bench exit $?"
tail -n 5 gpurun_out/r04l_bench.err | cut -c1-600
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04l_bench_search_legs.json") if l.startswith("{")][-1])
def pick(v, keys): return {k: v.get(k) for k in keys if k in v}
R = ["frac", "avg_launch_ms", "launches", "passes_per_launch", "traffic", "mfma_frac", "hbm_frac_of_needed_bytes", "unpaired_launch", "tail_pass", "power"]
print(json.dumps({"value": d["value"], "ms_per_step": d["ms_per_step"], "parity": d["parity_check"], "roofline": pick(d["roofline"], R),
  "kernel_ms_per_step": d.get("kernel_ms_per_step"),
  "gate": pick(d["full_list_gate"] or {}, ["queries", "query_indices", "ids_and_fp32_scores_bit_exact"]),
  "config5": pick(d.get("config5") or {}, ["queries_per_s", "parity_check", "full_list_gate"]), "config5_roofline": pick((d.get("config5") or {}).get("roofline") or {}, R),
  "real_size": pick(d.get("real_size") or {}, ["queries_per_s", "parity_check"]), "real_size_roofline": pick((d.get("real_size") or {}).get("roofline") or {}, R)}, indent=1)[:9000])
PY
