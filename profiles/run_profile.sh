#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline numbers are checked against.
#   profiles/run_profile.sh <tag>         (run on the GPU box from the repo root, e.g. via gpurun)
# Writes gpurun_out/prof_<tag>/... ; copy the *_kernel_stats.csv summary into profiles/ afterwards.
# Kernel trace / stats and PMC counters are collected in SEPARATE runs (never --pmc with a trace).
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="python $REPO/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-real-size --full-list-queries 8"   # (the real-size leg launches the headline kernel on 24.85 M rows: it would blur the per-kernel average of the 21 M launches)
BENCH_PMC="$BENCH --no-encoder --no-stage --no-certificate-leg --no-larger-k"   # the counter passes of the scan kernels (dense and sparse) do not need the encoder legs
BENCH_ENC="$BENCH --no-other-kernels --no-larger-k --no-config5 --no-certificate-leg --no-stage --no-splade --encode-stage-passages 0"
echo "== kernel trace + stats" 
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- $BENCH > "$OUT/trace.log" 2>&1
echo "exit $?" >> "$OUT/trace.log"
if [ "${2:-}" = "pmc" ]; then
  echo "== PMC pass 1: FETCH_SIZE"
  timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "bh_scan|bh_csr_scan" --output-format csv -d "$OUT/pmc_fetch" -o bench -- $BENCH_PMC --steps 1 > "$OUT/pmc_fetch.log" 2>&1
  echo "== PMC pass 2: WRITE_SIZE"
  timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "bh_scan|bh_csr_scan" --output-format csv -d "$OUT/pmc_write" -o bench -- $BENCH_PMC --steps 1 > "$OUT/pmc_write.log" 2>&1
  echo "== PMC pass 3: SQ"
  timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-include-regex "bh_scan|bh_csr_scan" --output-format csv -d "$OUT/pmc_sq" -o bench -- $BENCH_PMC --steps 1 > "$OUT/pmc_sq.log" 2>&1
  echo "== PMC pass 4: SQ counters of the encoder GEMM / attention kernels"
  timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-include-regex "bh_gemm|bh_attention" --output-format csv -d "$OUT/pmc_enc" -o bench -- $BENCH_ENC --steps 1 > "$OUT/pmc_enc.log" 2>&1
  python $REPO/profiles/summarize_pmc_enc.py "$OUT/pmc_enc" "$OUT/pmc_encoder_summary.json" > "$OUT/pmc_encoder_summary.log" 2>&1
fi
python $REPO/profiles/summarize_pmc.py "$OUT" "$OUT/pmc_summary.json" 21000000 768 21000000 30522 > "$OUT/pmc_summary.log" 2>&1
python $REPO/profiles/summarize_trace.py "$OUT/trace/bench_kernel_trace.csv" "$OUT" 4 12 > "$OUT/headline_from_trace.log" 2>&1
# keep only small artefacts for the copy-back
head -c 200000 "$OUT/trace/bench_kernel_trace.csv" > "$OUT/kernel_trace_head.csv" 2>/dev/null
cp "$OUT/trace/bench_kernel_stats.csv" "$OUT/kernel_stats.csv" 2>/dev/null
rm -rf "$OUT/trace" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_sq" "$OUT/pmc_enc"
ls -la "$OUT"
