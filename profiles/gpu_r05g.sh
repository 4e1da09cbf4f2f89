#!/bin/bash
# Round 5, evidence after the 16x16x32 encoder GEMM (gemm_f16_p16.h) and the transcendental-free GELU: the whole GPU suite, the bench line, a rocprofv3 kernel trace of the bench command, the PMC passes of the
# streaming kernels (FETCH_SIZE / WRITE_SIZE in separate runs, never with a trace), a kernel trace of the encoder alone.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_r05g
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
F="grep -v amdgpu.ids"
if [ "${1:-all}" != "noprofile" ]; then
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 600 --durations=10 2>&1 | $F | tail -30 | cut -c1-300 | tee gpurun_out/r05g_pytest_gpu.txt
fi
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05g_bench.json 2> gpurun_out/r05g_bench.err; echo "bench exit $?"
cd /tmp
BENCH="python $REPO/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-real-size --full-list-queries 8"
BENCH_PMC="$BENCH --steps 1 --no-encoder --no-stage --no-certificate-leg --no-larger-k --encode-stage-passages 0 --no-power-leg --full-list-queries 0 --splade-gate-queries 2"
echo "== kernel trace + stats of the bench command"
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- $BENCH > "$OUT/trace.log" 2>&1
echo "exit $?" >> "$OUT/trace.log"
grep '^{' "$OUT/trace.log" | tail -1 > "$REPO/gpurun_out/r05g_bench_under_rocprof.json"
cp "$OUT/trace/bench_kernel_stats.csv" "$REPO/gpurun_out/r05g_kernel_stats.csv" 2>/dev/null
rm -rf "$OUT/trace"
echo "== PMC pass 1: FETCH_SIZE"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "bh_scan|bh_csr_scan" --output-format csv -d "$OUT/pmc_fetch" -o bench -- $BENCH_PMC > "$OUT/pmc_fetch.log" 2>&1
echo "== PMC pass 2: WRITE_SIZE"
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "bh_scan|bh_csr_scan" --output-format csv -d "$OUT/pmc_write" -o bench -- $BENCH_PMC > "$OUT/pmc_write.log" 2>&1
python $REPO/profiles/summarize_pmc.py "$OUT" "$OUT/pmc_summary.json" 21000000 768 21000000 30522 > "$OUT/pmc_summary.log" 2>&1
cp "$OUT/pmc_summary.json" "$REPO/gpurun_out/r05g_pmc_summary.json" 2>/dev/null
cp "$OUT/hbm_traffic.json" "$REPO/gpurun_out/r05g_hbm_traffic.json" 2>/dev/null
rm -rf "$OUT/pmc_fetch" "$OUT/pmc_write"
echo "== kernel trace of the encoder alone (BERT-base, then the e5-large shape)"
for arch in bert e5_large; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/enc_$arch" -o enc -- python "$REPO/profiles/enc_trace.py" $arch 10 > "$OUT/enc_$arch.log" 2>&1
  find "$OUT/enc_$arch" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$REPO/gpurun_out/r05g_encoder_kernel_stats_$arch.csv"
  $F "$OUT/enc_$arch.log" | grep "forward ms" | tee -a "$REPO/gpurun_out/r05g_encoder_forward_ms.txt"
  rm -rf "$OUT/enc_$arch"
done
cd "$REPO"
python - <<'P'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r05g_bench.json") if l.startswith("{")][-1])
    r = d["roofline"]
    print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "parity_check")}))
    print(json.dumps({k: r.get(k) for k in ("avg_launch_ms", "launches", "frac", "frac_binding", "traffic", "avg_launch_ms_over_all_launches_of_this_kernel")}))
    print(json.dumps(r.get("secondary")))
except Exception as e:
    print("no bench line:", e)
P
head -6 gpurun_out/r05g_kernel_stats.csv | cut -c1-200
