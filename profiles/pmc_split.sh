#!/bin/bash
# FETCH_SIZE of the scan kernel with paired workgroups (query_split 2), nt on / off.  Run via gpurun from the repo root.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_split
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for NT in 1 0; do
  timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "bh_scan" --output-format csv -d "$OUT/nt$NT" -o bench -- \
    python $REPO/bench.py --gpus 1 --steps 1 --warmup 0 --no-cpu-baseline --no-encoder --query-split 2 --nontemporal $NT ${EXTRA:-} > "$OUT/nt$NT.log" 2>&1
  python - "$OUT/nt$NT" <<'PY'
import csv, glob, sys
vals = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == "FETCH_SIZE": vals.append(float(row["Counter_Value"]))
print(sys.argv[1], "launches", len(vals), "mean FETCH_SIZE*2048 GB:", [round(v * 2048 / 1e9, 2) for v in vals[:14]])
PY
  tail -c 400 "$OUT/nt$NT.log"
  rm -rf "$OUT/nt$NT"
done
