#!/bin/bash
# Round 4, last pass, call 2 of 2 (paired launches on by default): rocprofv3 kernel trace + stats of the bench command, the PMC
# passes of the 256-query scan kernel alone (FETCH_SIZE, WRITE_SIZE: separate runs, never with a trace), the same-box A/B of pair256.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_r04m
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="python $REPO/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-real-size --full-list-queries 8"
# counter passes: only the legs that launch bh_scan_topk256_kernel at full size (headline d = 768, configs[4] d = 1024)
BENCH_PMC="$BENCH --steps 1 --no-encoder --no-stage --no-certificate-leg --no-larger-k --no-other-kernels --no-splade --encode-stage-passages 0 --no-power-leg --full-list-queries 0"
echo "== kernel trace + stats"
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- $BENCH > "$OUT/trace.log" 2>&1
echo "exit $?" >> "$OUT/trace.log"
grep '^{' "$OUT/trace.log" | tail -1 > "$REPO/gpurun_out/r04m_bench_under_rocprof.json"
python $REPO/profiles/summarize_trace.py "$OUT/trace/bench_kernel_trace.csv" "$OUT" 4 12 > "$OUT/headline_from_trace.log" 2>&1
cp "$OUT/trace/bench_kernel_stats.csv" "$OUT/kernel_stats.csv" 2>/dev/null
rm -rf "$OUT/trace"
echo "== same-box A/B: pair256 off / on (headline geometry, all 21 M rows)"
cd "$REPO"
AB_CONFIGS="0,1;1,1" timeout 150 python profiles/ab_pair256.py 1 768 50 2>&1 | grep '^{' | tee "$REPO/gpurun_out/r04m_ab_pair256.jsonl"
echo "== host-side gap of a headline step"
timeout 120 python profiles/step_gap.py 2>&1 | grep -v amdgpu.ids | tee "$REPO/gpurun_out/r04m_step_gap.txt"
cd /tmp
echo "== PMC pass 1: FETCH_SIZE"
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "bh_scan_topk256" --output-format csv -d "$OUT/pmc_fetch" -o bench -- $BENCH_PMC > "$OUT/pmc_fetch.log" 2>&1
echo "== PMC pass 2: WRITE_SIZE"
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "bh_scan_topk256" --output-format csv -d "$OUT/pmc_write" -o bench -- $BENCH_PMC > "$OUT/pmc_write.log" 2>&1
python $REPO/profiles/summarize_pmc.py "$OUT" "$OUT/pmc_summary.json" 21000000 768 > "$OUT/pmc_summary.log" 2>&1
rm -rf "$OUT/pmc_fetch" "$OUT/pmc_write"
tail -c 1500 "$OUT/headline_from_trace.log"
cat "$OUT/hbm_traffic.json" 2>/dev/null | head -60
ls -la "$OUT"
