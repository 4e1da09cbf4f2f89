#!/bin/bash
# the whole bench line on the final build (paired launches, host pools sized to the CPU quota)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04o_bench.json 2> gpurun_out/r04o_bench.err
echo "bench exit $?"
tail -n 3 gpurun_out/r04o_bench.err | cut -c1-400
cat /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo; cat /proc/loadavg
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04o_bench.json") if l.startswith("{")][-1])
def pick(v, keys): return {k: v.get(k) for k in keys if k in v}
R = ["frac", "avg_launch_ms", "launches", "passes_per_launch", "traffic", "mfma_frac", "hbm_frac_of_needed_bytes"]
print(json.dumps({"value": d["value"], "ms_per_step": d["ms_per_step"], "parity": d["parity_check"], "host": d.get("host"), "roofline": pick(d["roofline"], R),
  "kernel_ms_per_step": d.get("kernel_ms_per_step"), "gate": pick(d["full_list_gate"] or {}, ["queries", "ids_and_fp32_scores_bit_exact"]),
  "cpu_baseline": pick(d.get("cpu_baseline") or {}, ["value", "cores", "host_logical_cpus"]),
  "passages_per_s": d.get("passages_per_s"), "encoder_frac": (d.get("encoder_roofline") or {}).get("frac"),
  "config5": pick(d.get("config5") or {}, ["queries_per_s", "parity_check"]), "config5_roofline": pick((d.get("config5") or {}).get("roofline") or {}, R),
  "real_size": pick(d.get("real_size") or {}, ["queries_per_s", "parity_check"]),
  "certificate": pick(d.get("certificate") or {}, ["queries_per_s", "fallback_ms", "parity_check"]),
  "encode_stage": {k: (pick(v, ["passages_per_s", "steady_state_passages_per_s"]) if isinstance(v, dict) else v) for k, v in (d.get("encode_stage") or {}).items() if k != "workload"},
  "rerank": {k: (pick(v, ["pairs_per_s"]) if isinstance(v, dict) else v) for k, v in (d.get("rerank") or {}).items()},
  "splade_search": pick(d.get("splade_search") or {}, ["queries_per_s", "parity_check"]), "splade_frac": ((d.get("splade_search") or {}).get("roofline") or {}).get("frac"),
  "splade_encode": pick(d.get("splade_encode") or {}, ["passages_per_s"]), "stage_full": pick(d.get("retrieve_stage_full") or {}, ["queries_per_s", "seconds", "error"])}, indent=1)[:7000])
PY
