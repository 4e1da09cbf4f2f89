#!/bin/bash
# Round 5, third GPU topic: the balanced remainder (index.hip option balance_tail) with idle waves (scan_topk256.hip BhScanArgs::nq_valid):
# parity of every search test, then the same-box A/B at the headline size and on an eighth of the corpus, then the bench line.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
export TMPDIR=/tmp
F="grep -v amdgpu.ids"
timeout 900 python -m pytest tests/test_gpu_search.py tests/test_gpu_abi.py tests/test_gpu_retrieve.py tests/test_gpu_nccl.py -m gpu -q -x --tb=short -p no:cacheprovider --timeout 400 --durations=6 \
    2>&1 | $F | tail -25 | cut -c1-400 | tee gpurun_out/r05c_pytest_search.txt
timeout 200 python profiles/ab_balance_tail.py 1 2>&1 | $F | tee gpurun_out/r05c_ab_balance_tail.jsonl
timeout 120 python profiles/ab_balance_tail.py 8 2>&1 | $F | tee -a gpurun_out/r05c_ab_balance_tail.jsonl
timeout 200 python profiles/ab_balance_tail.py 1 1024 200 2>&1 | $F | tee -a gpurun_out/r05c_ab_balance_tail.jsonl
timeout 300 python bench.py --gpus 1 --steps 10 --warmup 2 --no-encoder --no-splade --no-cpu-baseline --no-real-size --no-certificate-leg --no-larger-k --no-other-kernels > gpurun_out/r05c_bench_search.json 2> gpurun_out/r05c_bench.err; echo "bench exit $?"
python - <<'P'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r05c_bench_search.json") if l.startswith("{")][-1])
    r = d["roofline"]
    print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "parity_check")}), json.dumps({k: r.get(k) for k in ("avg_launch_ms", "launches", "frac", "frac_binding", "balanced_launch", "unpaired_launch", "tail_pass")}))
    print(json.dumps(d.get("full_list_gate"))[:600])
    print(json.dumps(d.get("config5", {}).get("queries_per_s")), json.dumps(d.get("config5", {}).get("roofline", {}).get("balanced_launch")))
except Exception as e:
    print("no bench line:", e)
P
