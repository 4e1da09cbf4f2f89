#!/bin/bash
# final build (paired launches): the one-GPU proxy of the 8-GPU case for both geometries — pairing on and off on the same box —, then the
# encoder legs of the bench (BERT-base and the new NomicBert leg)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
timeout 100 python profiles/shard_sweep.py 768 50 2837 1 8 2>/dev/null > gpurun_out/r04q_shard_sweep.json; echo "sweep768 exit $?"
timeout 100 python profiles/shard_sweep.py 768 50 2837 1 8 pair256=0 2>/dev/null > gpurun_out/r04q_shard_sweep_unpaired.json; echo "sweep768 unpaired exit $?"
timeout 100 python profiles/shard_sweep.py 1024 200 1000 1 8 2>/dev/null > gpurun_out/r04q_shard_sweep_d1024.json; echo "sweep1024 exit $?"
timeout 100 python profiles/shard_sweep.py 1024 200 1000 1 8 pair256=0 2>/dev/null > gpurun_out/r04q_shard_sweep_d1024_unpaired.json; echo "sweep1024 unpaired exit $?"
python - <<'PY'
import json
for f in ("r04q_shard_sweep", "r04q_shard_sweep_unpaired", "r04q_shard_sweep_d1024", "r04q_shard_sweep_d1024_unpaired"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
        print(f, [(s["g"], round(s["wall_ms"], 2), round(s["scan_ms_per_pass"], 3), round(s["speedup_vs_full_corpus"], 2)) for s in d["shards"]])
    except Exception as e:
        print(f, "failed", e)
PY
timeout 200 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-real-size --no-config5 --no-certificate-leg --no-larger-k --no-other-kernels --no-splade --no-stage --encode-stage-passages 0 --full-list-queries 0 --no-power-leg > gpurun_out/r04q_bench_encoder_legs.json 2> gpurun_out/r04q_bench.err; echo "bench exit $?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04q_bench_encoder_legs.json") if l.startswith("{")][-1])
print(json.dumps({"passages_per_s": d.get("passages_per_s"), "encoder_frac": (d.get("encoder_roofline") or {}).get("frac"), "nomic_encode": d.get("nomic_encode"),
                  "encoder_error": d.get("encoder_error"), "rerank": {k: v.get("pairs_per_s") if isinstance(v, dict) else v for k, v in (d.get("rerank") or {}).items()}}, indent=1)[:3000])
PY
