#!/bin/bash
# Round 6, second GPU call: rerank pipeline tests + bench leg through Rerank.eval, the in-kernel ablation ladder of the paired scan
# instantiation, the whole-search 8-GPU proxy (gather + 8-list merge + broadcast included) for both geometries.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
F="grep -v amdgpu.ids"
timeout 600 python -m pytest tests/test_gpu_rerank.py tests/test_gpu_search.py -m gpu -q --tb=short -p no:cacheprovider --timeout 300 -x -k "not full_size" 2>&1 | $F | tail -8 | cut -c1-300 | tee gpurun_out/r06b_pytest.txt
timeout 300 python profiles/ablate_paired.py 2> gpurun_out/r06b_ablate.err > gpurun_out/r06_scan_paired_ablation.json; echo "ablate exit $?"; $F gpurun_out/r06b_ablate.err | tail -8 | cut -c1-250
timeout 300 python profiles/shard_search_proxy.py 768 50 2837 2> gpurun_out/r06b_proxy.err > gpurun_out/r06_shard_search_proxy.json; echo "proxy768 exit $?"
timeout 300 python profiles/shard_search_proxy.py 1024 200 1000 2>> gpurun_out/r06b_proxy.err > gpurun_out/r06_shard_search_proxy_d1024.json; echo "proxy1024 exit $?"
$F gpurun_out/r06b_proxy.err | tail -5 | cut -c1-300
python - <<'PY'
import json
for f in ("r06_shard_search_proxy", "r06_shard_search_proxy_d1024"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
        print(f, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.items() if k not in ("note", "workload", "pieces_ms")}, {k: round(v, 3) for k, v in d["pieces_ms"].items()})
    except Exception as e:
        print(f, "failed", e)
PY
timeout 400 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-real-size --no-config5 --no-certificate-leg --no-larger-k --no-other-kernels --no-splade --no-stage --encode-stage-passages 0 --full-list-queries 0 --no-power-leg > gpurun_out/r06b_bench_encoder_legs.json 2> gpurun_out/r06b_bench.err; echo "bench exit $?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r06b_bench_encoder_legs.json") if l.startswith("{")][-1])
print(json.dumps({"value": d.get("value"), "ms": d.get("ms_per_step"), "roofline_bound": d["roofline"].get("bound"), "mfma": d["roofline"].get("mfma"), "backend": d["config"].get("backend"),
                  "passages_per_s": d.get("passages_per_s"), "encoder_roofline": d.get("encoder_roofline")}, indent=None)[:1500])
for name, v in (d.get("rerank") or {}).items():
    if isinstance(v, dict):
        print(name, round(v.get("pairs_per_s", 0)), round(v["roofline"]["frac"], 3), json.dumps(v.get("through_rerank_eval"))[:900])
PY
tail -3 gpurun_out/r06b_bench.err | cut -c1-300
