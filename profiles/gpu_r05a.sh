#!/bin/bash
# Round 5, first GPU call: the new parity tests (SPLADE at configs[3] size against the oracle, the RCCL transport at world size 1, the
# hub form of NomicBert), the store paths round 4 left behind an option, a kernel trace of the encoder AS BUILT NOW, one bench line.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
export TMPDIR=/tmp
F="grep -v amdgpu.ids"
timeout 400 python -m pytest tests/test_gpu_nccl.py "tests/test_gpu_sparse.py::test_full_size_sparse" tests/test_gpu_nomic.py tests/test_gpu_store_paths.py \
    -m gpu -q --tb=short -p no:cacheprovider --timeout 300 --durations=8 2>&1 | $F | tail -40 | cut -c1-400 | tee gpurun_out/r05a_pytest_new.txt
timeout 90 python profiles/enc_ab_option.py gemm_full_line_stores 1 2 2>&1 | $F | tee gpurun_out/r05a_ab_full_line_level2.txt
ENC_ARCH=nomic timeout 90 python profiles/enc_ab_option.py gemm_full_line_stores 1 2 2>&1 | $F | tee -a gpurun_out/r05a_ab_full_line_level2.txt
# (this call also A/B-ed a static s_setprio for waves 4-7 of the persistent GEMM — 14.352 vs 14.366 ms, nothing: profiles/r05a_ab_static_prio.txt; the knob was removed)
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_r05a_enc" -o enc -- python "$REPO/profiles/enc_trace.py" bert 10 > "$REPO/gpurun_out/r05a_enc_trace.log" 2>&1; echo "rocprof exit $?")
find gpurun_out/prof_r05a_enc -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r05a_encoder_kernel_stats.csv
rm -rf gpurun_out/prof_r05a_enc
$F gpurun_out/r05a_enc_trace.log | tail -3
cut -c1-160 gpurun_out/r05a_encoder_kernel_stats.csv | head -24
timeout 420 python bench.py --gpus 1 --steps 10 --warmup 2 > gpurun_out/r05a_bench.json 2> gpurun_out/r05a_bench.err; echo "bench exit $?"
tail -c 1500 gpurun_out/r05a_bench.err | $F | tail -5
python - <<'P'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r05a_bench.json") if l.startswith("{")][-1])
    print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "parity_check")}))
    print(json.dumps(d["roofline"]["secondary"]))
    print(json.dumps(d.get("splade_search"))[:1500])
except Exception as e:
    print("no bench line:", e)
P
