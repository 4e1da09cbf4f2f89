#!/bin/bash
# Round 6, first GPU call: (1) the new SPLADE-head tests (DistilBERT / RoBERTa masked-LM heads), (2) SQ counters (MFMA busy cycles, wait
# cycles, LDS conflicts; GRBM_GUI_ACTIVE for the effective clock) of the PAIRED scan instantiation and of the p16 encoder GEMM — their own
# rocprofv3 --pmc passes, never with a trace —, (3) the one-GPU shard proxy on this build.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_r06a
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
F="grep -v amdgpu.ids"
timeout 600 python -m pytest tests/test_gpu_splade.py tests/test_gpu_hf_path.py -m gpu -q --tb=short -p no:cacheprovider --timeout 300 2>&1 | $F | tail -15 | cut -c1-300 | tee gpurun_out/r06a_pytest_splade_heads.txt
cd /tmp
BENCH_PMC="python $REPO/bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --no-real-size --no-encoder --no-stage --no-certificate-leg --no-larger-k --encode-stage-passages 0 --no-power-leg --full-list-queries 0 --no-splade --no-config5 --no-other-kernels"
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY"
SQ2="GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"
echo "== SQ pass 1 (scan)"
timeout 400 rocprofv3 --pmc $SQ1 --kernel-include-regex "bh_scan_topk256" --output-format csv -d "$OUT/sq1_scan" -o bench -- $BENCH_PMC > "$OUT/sq1_scan.log" 2>&1; echo "exit $?"
echo "== SQ pass 2 (scan)"
timeout 400 rocprofv3 --pmc $SQ2 --kernel-include-regex "bh_scan_topk256" --output-format csv -d "$OUT/sq2_scan" -o bench -- $BENCH_PMC > "$OUT/sq2_scan.log" 2>&1; echo "exit $?"
echo "== SQ pass 1 (encoder)"
timeout 300 rocprofv3 --pmc $SQ1 --kernel-include-regex "bh_gemm|bh_attention|bh_layernorm" --output-format csv -d "$OUT/sq1_enc" -o enc -- python "$REPO/profiles/enc_trace.py" bert 3 > "$OUT/sq1_enc.log" 2>&1; echo "exit $?"
echo "== SQ pass 2 (encoder)"
timeout 300 rocprofv3 --pmc $SQ2 --kernel-include-regex "bh_gemm|bh_attention|bh_layernorm" --output-format csv -d "$OUT/sq2_enc" -o enc -- python "$REPO/profiles/enc_trace.py" bert 3 > "$OUT/sq2_enc.log" 2>&1; echo "exit $?"
tail -3 "$OUT/sq2_scan.log" | cut -c1-300
tail -2 "$OUT/sq2_enc.log" | cut -c1-300
python "$REPO/profiles/summarize_sq.py" "$OUT" "$REPO/gpurun_out/r06a_sq_summary.json" 2>&1 | tail -40
rm -rf "$OUT"/sq1_scan "$OUT"/sq2_scan "$OUT"/sq1_enc "$OUT"/sq2_enc
cd "$REPO"
echo "== shard proxy"
timeout 150 python profiles/shard_sweep.py 768 50 2837 1 8 2>/dev/null > gpurun_out/r06a_shard_sweep.json; echo "sweep768 exit $?"
timeout 150 python profiles/shard_sweep.py 1024 200 1000 1 8 2>/dev/null > gpurun_out/r06a_shard_sweep_d1024.json; echo "sweep1024 exit $?"
python - <<'PY'
import json
for f in ("r06a_shard_sweep", "r06a_shard_sweep_d1024"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
        print(f, [(s["g"], round(s["wall_ms"], 2), round(s["scan_ms_per_pass"], 3), round(s["speedup_vs_full_corpus"], 2)) for s in d["shards"]])
    except Exception as e:
        print(f, "failed", e)
PY
