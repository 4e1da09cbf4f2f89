#!/bin/bash
# Round 6, third GPU call: gte ("new") encoder tests, the merge kernel's ranking paths, the whole-search proxies after the merge fix,
# the host gap of a rerank launch.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
F="grep -v amdgpu.ids"
timeout 900 python -m pytest tests/test_gpu_gte.py tests/test_gpu_nomic.py tests/test_gpu_encoder.py tests/test_gpu_store_paths.py -m gpu -q --tb=short -p no:cacheprovider --timeout 300 2>&1 | $F | tail -25 | cut -c1-400 | tee gpurun_out/r06c_pytest_gte.txt
timeout 600 python -m pytest tests/test_gpu_search.py -m gpu -q --tb=short -p no:cacheprovider --timeout 300 -k "merge or shard" 2>&1 | $F | tail -8 | cut -c1-300 | tee gpurun_out/r06c_pytest_merge.txt
timeout 300 python profiles/shard_search_proxy.py 768 50 2837 out=gpurun_out/r06_shard_search_proxy.json > /dev/null 2> gpurun_out/r06c_proxy.err; echo "proxy768 exit $?"
timeout 300 python profiles/shard_search_proxy.py 1024 200 1000 out=gpurun_out/r06_shard_search_proxy_d1024.json > /dev/null 2>> gpurun_out/r06c_proxy.err; echo "proxy1024 exit $?"
python - <<'PY'
import json
for f in ("r06_shard_search_proxy", "r06_shard_search_proxy_d1024"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
        print(f, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.items() if k not in ("note", "workload", "pieces_ms")}, {k: round(v, 3) for k, v in d["pieces_ms"].items()})
    except Exception as e:
        print(f, "failed", e)
PY
timeout 300 python profiles/rerank_gap.py 2>&1 | $F | tail -6
