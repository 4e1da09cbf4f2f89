#!/bin/bash
# NomicBert: the gated fold in the GEMM epilogue (BH_EPI_SWIGLU) — parity, then fused vs unfused on the bench batch, then the bench's encoder legs
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_gpu_nomic.py "tests/test_gpu_encoder.py::test_encoder_matches_hf_golden_fixture" "tests/test_gpu_encoder.py::test_gemm_persistent_many_tiles_per_block" tests/test_gpu_splade.py -m gpu -q --tb=short -p no:cacheprovider -s --timeout 120 2>&1 | grep -v "amdgpu.ids\|Writing model\|Loading weights\|Encoding:\|Retrieving\|Load sparse" | tail -25 | cut -c1-300 | tee gpurun_out/r04s_pytest_nomic.txt
ENC_ARCH=nomic timeout 120 python profiles/enc_ab_option.py ffn_fused 0 1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04s_ab_ffn_fused.txt
timeout 200 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-real-size --no-config5 --no-certificate-leg --no-larger-k --no-other-kernels --no-splade --no-stage --encode-stage-passages 0 --full-list-queries 0 --no-power-leg > gpurun_out/r04s_bench_encoder_legs.json 2> gpurun_out/r04s_bench.err; echo "bench exit $?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04s_bench_encoder_legs.json") if l.startswith("{")][-1])
n = d.get("nomic_encode") or {}
print(json.dumps({"passages_per_s": d.get("passages_per_s"), "encoder_frac": (d.get("encoder_roofline") or {}).get("frac"),
                  "nomic": {k: n.get(k) for k in ("passages_per_s", "ms_per_step_kernels", "finite_and_shaped")}, "nomic_frac": (n.get("roofline") or {}).get("frac"),
                  "encoder_error": d.get("encoder_error")}, indent=1))
PY
