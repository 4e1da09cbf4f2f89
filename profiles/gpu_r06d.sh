#!/bin/bash
# Round 6, fourth GPU call: the whole GPU suite on the build with the GELU-gated fold / ALiBi / binary-search merge, then the proxies and
# the rerank legs again.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
F="grep -v amdgpu.ids"
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 --durations=8 2>&1 | $F | tail -40 | cut -c1-400 | tee gpurun_out/r06d_pytest_gpu.txt
timeout 300 python profiles/shard_search_proxy.py 768 50 2837 out=gpurun_out/r06_shard_search_proxy.json > /dev/null 2> gpurun_out/r06d_proxy.err; echo "proxy768 exit $?"
timeout 300 python profiles/shard_search_proxy.py 1024 200 1000 out=gpurun_out/r06_shard_search_proxy_d1024.json > /dev/null 2>> gpurun_out/r06d_proxy.err; echo "proxy1024 exit $?"
python - <<'PY'
import json
for f in ("r06_shard_search_proxy", "r06_shard_search_proxy_d1024"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
        print(f, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.items() if k not in ("note", "workload", "pieces_ms")}, {k: round(v, 3) for k, v in d["pieces_ms"].items()})
    except Exception as e:
        print(f, "failed", e)
PY
timeout 400 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-real-size --no-config5 --no-certificate-leg --no-larger-k --no-other-kernels --no-splade --no-stage --encode-stage-passages 0 --full-list-queries 0 --no-power-leg > gpurun_out/r06d_bench_encoder_legs.json 2> gpurun_out/r06d_bench.err; echo "bench exit $?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r06d_bench_encoder_legs.json") if l.startswith("{")][-1])
print(json.dumps({"value": d.get("value"), "ms": d.get("ms_per_step"), "passages_per_s": d.get("passages_per_s"), "enc_frac": d.get("encoder_roofline", {}).get("frac")}))
for name, v in (d.get("rerank") or {}).items():
    if isinstance(v, dict):
        t = v.get("through_rerank_eval") or {}
        print(name, round(v.get("pairs_per_s", 0)), round(v["roofline"]["frac"], 3), {k: (round(t[k]["pairs_per_s"]), round(t[k]["roofline"]["frac"], 3), round(t[k]["roofline"]["frac_kernels_only"], 3), t[k]["launches"]) for k in ("coalesced_256", "one_launch_per_yaml_batch") if k in t}, t.get("identical_to_per_batch_loop"), t.get("error"))
PY
