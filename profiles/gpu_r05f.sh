#!/bin/bash
# round 5: kernel trace of the encoder alone with the 16x16x32 GEMM as default
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT; OUT=/tmp/prof_r05f; mkdir -p $OUT $REPO/gpurun_out
for arch in bert e5_large; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/enc_$arch" -o enc -- python "$REPO/profiles/enc_trace.py" $arch 10 > "$OUT/enc_$arch.log" 2>&1
  find "$OUT/enc_$arch" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$REPO/gpurun_out/r05f_encoder_kernel_stats_$arch.csv"
  grep "forward ms" "$OUT/enc_$arch.log" | tee -a "$REPO/gpurun_out/r05f_encoder_forward_ms.txt"
done
