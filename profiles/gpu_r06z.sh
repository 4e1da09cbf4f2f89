#!/bin/bash
# Round 6, final evidence: the whole GPU suite, the bench line, a rocprofv3 kernel trace of the bench command, the PMC passes of the
# streaming kernels (FETCH_SIZE / WRITE_SIZE in separate runs, never with a trace), a kernel trace of the encoder alone.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_r06z
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
F="grep -v amdgpu.ids"
if [ "${1:-all}" != "noprofile" ]; then
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 --durations=10 2>&1 | $F | tail -30 | cut -c1-300 | tee gpurun_out/r06z_pytest_gpu.txt
fi
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06z_bench.json 2> gpurun_out/r06z_bench.err; echo "bench exit $?"
cd /tmp
BENCH="python $REPO/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-real-size --full-list-queries 8"
BENCH_PMC="$BENCH --steps 1 --no-encoder --no-stage --no-certificate-leg --no-larger-k --encode-stage-passages 0 --no-power-leg --full-list-queries 0 --splade-gate-queries 2"
echo "== kernel trace + stats of the bench command"
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- $BENCH > "$OUT/trace.log" 2>&1
echo "exit $?" >> "$OUT/trace.log"
grep '^{' "$OUT/trace.log" | tail -1 > "$REPO/gpurun_out/r06z_bench_under_rocprof.json"
cp "$OUT/trace/bench_kernel_stats.csv" "$REPO/gpurun_out/r06z_kernel_stats.csv" 2>/dev/null
rm -rf "$OUT/trace"
echo "== PMC pass 1: FETCH_SIZE"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "bh_scan|bh_csr_scan" --output-format csv -d "$OUT/pmc_fetch" -o bench -- $BENCH_PMC > "$OUT/pmc_fetch.log" 2>&1
echo "== PMC pass 2: WRITE_SIZE"
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "bh_scan|bh_csr_scan" --output-format csv -d "$OUT/pmc_write" -o bench -- $BENCH_PMC > "$OUT/pmc_write.log" 2>&1
python $REPO/profiles/summarize_pmc.py "$OUT" "$OUT/pmc_summary.json" 21000000 768 21000000 30522 > "$OUT/pmc_summary.log" 2>&1
cp "$OUT/pmc_summary.json" "$REPO/gpurun_out/r06z_pmc_summary.json" 2>/dev/null
cp "$OUT/hbm_traffic.json" "$REPO/gpurun_out/r06z_hbm_traffic.json" 2>/dev/null
rm -rf "$OUT/pmc_fetch" "$OUT/pmc_write"
echo "== SQ counters (two passes of 8, kernel-filtered, no trace): the scan kernels of the bench command, the encoder-only command"
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY"
SQ2="GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"
BENCH_SQ="$BENCH_PMC --no-splade --no-config5 --no-other-kernels"
mkdir -p "$OUT/sq"
timeout 400 rocprofv3 --pmc $SQ1 --kernel-include-regex "bh_scan_topk256" --output-format csv -d "$OUT/sq/sq1_scan" -o bench -- $BENCH_SQ > "$OUT/sq1_scan.log" 2>&1; echo "sq1 scan exit $?"
timeout 400 rocprofv3 --pmc $SQ2 --kernel-include-regex "bh_scan_topk256" --output-format csv -d "$OUT/sq/sq2_scan" -o bench -- $BENCH_SQ > "$OUT/sq2_scan.log" 2>&1; echo "sq2 scan exit $?"
timeout 300 rocprofv3 --pmc $SQ1 --kernel-include-regex "bh_gemm|bh_attention|bh_layernorm" --output-format csv -d "$OUT/sq/sq1_enc" -o enc -- python "$REPO/profiles/enc_trace.py" bert 3 > "$OUT/sq1_enc.log" 2>&1; echo "sq1 enc exit $?"
timeout 300 rocprofv3 --pmc $SQ2 --kernel-include-regex "bh_gemm|bh_attention|bh_layernorm" --output-format csv -d "$OUT/sq/sq2_enc" -o enc -- python "$REPO/profiles/enc_trace.py" bert 3 > "$OUT/sq2_enc.log" 2>&1; echo "sq2 enc exit $?"
python "$REPO/profiles/summarize_sq.py" "$OUT/sq" "$REPO/gpurun_out/r06z_sq_summary.json" 2>&1 | tail -12 | cut -c1-400
rm -rf "$OUT/sq"
echo "== kernel trace of the encoder alone (BERT-base, then the e5-large shape)"
for arch in bert e5_large; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/enc_$arch" -o enc -- python "$REPO/profiles/enc_trace.py" $arch 10 > "$OUT/enc_$arch.log" 2>&1
  find "$OUT/enc_$arch" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$REPO/gpurun_out/r06z_encoder_kernel_stats_$arch.csv"
  $F "$OUT/enc_$arch.log" | grep "forward ms" | tee -a "$REPO/gpurun_out/r06z_encoder_forward_ms.txt"
  rm -rf "$OUT/enc_$arch"
done
cd "$REPO"
echo "== one-GPU proxies of the 8-GPU case: scan-only sweep, whole ShardedSearcher.search"
timeout 150 python profiles/shard_sweep.py 768 50 2837 1 8 2>/dev/null > gpurun_out/r06z_shard_sweep.json; echo "sweep768 exit $?"
timeout 150 python profiles/shard_sweep.py 1024 200 1000 1 8 2>/dev/null > gpurun_out/r06z_shard_sweep_d1024.json; echo "sweep1024 exit $?"
timeout 300 python profiles/shard_search_proxy.py 768 50 2837 out=gpurun_out/r06z_shard_search_proxy.json > /dev/null 2>&1; echo "proxy768 exit $?"
timeout 300 python profiles/shard_search_proxy.py 1024 200 1000 out=gpurun_out/r06z_shard_search_proxy_d1024.json > /dev/null 2>&1; echo "proxy1024 exit $?"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | $F | tail -2 | tee gpurun_out/r06z_smoke.txt
python - <<'P'
import json
for f in ("r06z_shard_search_proxy", "r06z_shard_search_proxy_d1024"):
    try:
        d = json.load(open(f"gpurun_out/{f}.json"))
        print(f, round(d["projected_speedup_whole_search"], 2), round(d["scan_only_speedup_for_comparison"], 2), {k: round(v, 3) for k, v in d["pieces_ms"].items()})
    except Exception as e:
        print(f, "failed", e)
try:
    d = json.loads([l for l in open("gpurun_out/r06z_bench.json") if l.startswith("{")][-1])
    r = d["roofline"]
    print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "parity_check")}))
    print(json.dumps({k: r.get(k) for k in ("avg_launch_ms", "launches", "frac", "frac_binding", "traffic", "avg_launch_ms_over_all_launches_of_this_kernel")}))
    print(json.dumps(r.get("secondary")))
except Exception as e:
    print("no bench line:", e)
P
head -6 gpurun_out/r06z_kernel_stats.csv | cut -c1-200
