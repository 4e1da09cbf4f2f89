#!/bin/bash
# full-line stores on by default: every GPU test that runs the encoder kernels, then the whole bench line
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
timeout 175 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_nomic.py tests/test_gpu_splade.py tests/test_gpu_rerank.py tests/test_gpu_deberta.py tests/test_gpu_hf_path.py tests/test_gpu_ut1.py -m gpu -q -x --tb=short -p no:cacheprovider --timeout 100 2>&1 | grep -v "amdgpu.ids\|Writing model\|Loading weights\|Encoding:\|Retrieving\|Load sparse\|torch_dtype" | tail -12 | cut -c1-300 | tee gpurun_out/r04u_pytest_encoder_tests.txt
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04u_bench.json 2> gpurun_out/r04u_bench.err
echo "bench exit $?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04u_bench.json") if l.startswith("{")][-1])
def pick(v, keys): return {k: v.get(k) for k in keys if k in v}
n = d.get("nomic_encode") or {}
print(json.dumps({"value": d["value"], "ms_per_step": d["ms_per_step"], "parity": d["parity_check"], "frac": d["roofline"]["frac"], "traffic": d["roofline"]["traffic"],
  "cpu_baseline": pick(d.get("cpu_baseline") or {}, ["value", "cores", "measured_seconds"]),
  "passages_per_s": d.get("passages_per_s"), "encoder_frac": (d.get("encoder_roofline") or {}).get("frac"), "encoder_ms": (d.get("encoder") or {}).get("ms_per_step_kernels"),
  "nomic": pick(n, ["passages_per_s", "ms_per_step_kernels"]), "nomic_frac": (n.get("roofline") or {}).get("frac"),
  "encode_stage": {k: (pick(v, ["passages_per_s", "steady_state_passages_per_s"]) if isinstance(v, dict) else v) for k, v in (d.get("encode_stage") or {}).items() if k != "workload"},
  "rerank": {k: (v.get("pairs_per_s") if isinstance(v, dict) else v) for k, v in (d.get("rerank") or {}).items()},
  "splade_encode": pick(d.get("splade_encode") or {}, ["passages_per_s"]), "splade_search": pick(d.get("splade_search") or {}, ["queries_per_s", "parity_check"]),
  "config5": pick(d.get("config5") or {}, ["queries_per_s", "parity_check"]), "real_size": pick(d.get("real_size") or {}, ["queries_per_s", "parity_check"]),
  "certificate": pick(d.get("certificate") or {}, ["queries_per_s", "parity_check"])}, indent=1)[:5000])
PY
