#!/bin/bash
# Round 4, first GPU pass: parity tests (new full-size pins), headline bench, vendor GEMM yardstick, L2->LDS load-path ubench.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
mkdir -p gpurun_out
echo "== pytest -m gpu"; date
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=15 > gpurun_out/r04a_pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a gpurun_out/r04a_pytest_gpu.log
tail -n 45 gpurun_out/r04a_pytest_gpu.log
echo "== ubench L2->LDS"; date
timeout 120 profiles/bin/ubench_l2_lds > gpurun_out/r04a_ubench_l2_lds.jsonl 2>&1; cat gpurun_out/r04a_ubench_l2_lds.jsonl
timeout 120 profiles/bin/ubench_l2_lds 68608 768 3072 > gpurun_out/r04a_ubench_l2_lds_ffn2.jsonl 2>&1; cat gpurun_out/r04a_ubench_l2_lds_ffn2.jsonl
echo "== vendor GEMM yardstick"; date
timeout 300 python profiles/gemm_yardstick.py gpurun_out/r04a_gemm_yardstick.json 2>&1 | tail -n 12
echo "== bench.py"; date
timeout 900 python bench.py --gpus 1 --steps 5 --warmup 2 > gpurun_out/r04a_bench.json 2> gpurun_out/r04a_bench.err
echo "bench exit $?" | tee -a gpurun_out/r04a_bench.err
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r04a_bench.json") if l.startswith("{")][-1])
    keep = {k: d.get(k) for k in ("value", "ms_per_step", "parity_check", "full_list_gate", "passages_per_s", "uncertified_queries")}
    keep["roofline_frac"] = d["roofline"]["frac"]
    for leg in ("config5", "real_size", "certificate", "encoder_roofline", "splade_search"):
        v = d.get(leg)
        if isinstance(v, dict):
            keep[leg] = {k: v.get(k) for k in ("queries_per_s", "ms_per_step", "parity_check", "full_list_gate", "frac", "error", "finalize_seconds") if k in v}
            if "roofline" in v: keep[leg]["frac"] = v["roofline"].get("frac")
    print(json.dumps(keep, indent=1)[:6000])
except Exception as e:
    print("no bench line:", e)
PY
tail -n 5 gpurun_out/r04a_bench.err
date
