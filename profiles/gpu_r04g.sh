#!/bin/bash
# Round 4, full pass on the final build: every GPU test, the whole bench line.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=8 > gpurun_out/r04g_pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a gpurun_out/r04g_pytest_gpu.log
tail -n 16 gpurun_out/r04g_pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04g_bench.json 2> gpurun_out/r04g_bench.err
echo "bench exit $?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04g_bench.json") if l.startswith("{")][-1])
def pick(v, keys): return {k: v.get(k) for k in keys if k in v}
print(json.dumps({"value": d["value"], "ms_per_step": d["ms_per_step"], "parity": d["parity_check"], "roofline": pick(d["roofline"], ["frac", "avg_launch_ms", "traffic", "mfma_frac"]),
  "gate": pick(d["full_list_gate"] or {}, ["queries", "ids_and_fp32_scores_bit_exact"]), "cpu_baseline": d.get("cpu_baseline"),
  "passages_per_s": d.get("passages_per_s"), "encoder_frac": (d.get("encoder_roofline") or {}).get("frac"),
  "config5": pick(d.get("config5") or {}, ["queries_per_s", "parity_check"]), "config5_frac": ((d.get("config5") or {}).get("roofline") or {}).get("frac"),
  "real_size": pick(d.get("real_size") or {}, ["queries_per_s", "parity_check"]),
  "certificate": pick(d.get("certificate") or {}, ["queries_per_s", "fallback_ms", "fallback_filter_passes", "parity_check"]),
  "encode_stage": {k: (pick(v, ["passages_per_s", "steady_state_passages_per_s", "first_batch_seconds", "last_chunk_write_seconds"]) if isinstance(v, dict) else v) for k, v in (d.get("encode_stage") or {}).items() if k != "workload"},
  "rerank": d.get("rerank"), "larger_k": d.get("larger_k"),
  "splade_search": pick(d.get("splade_search") or {}, ["queries_per_s", "parity_check"]), "splade_frac": ((d.get("splade_search") or {}).get("roofline") or {}).get("frac"),
  "splade_encode": pick(d.get("splade_encode") or {}, ["passages_per_s"]), "stage_full": pick(d.get("retrieve_stage_full") or {}, ["queries_per_s", "seconds"])}, indent=1)[:9000])
PY
