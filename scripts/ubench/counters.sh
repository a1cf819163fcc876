#!/bin/bash
# Counter calibration: build here, run on the GPU box.
#   bash scripts/ubench/counters.sh build        (hipcc, no GPU needed)
#   bash scripts/ubench/counters.sh run OUTDIR   (three rocprofv3 passes + the table)
cd "$(dirname "$0")/../.."
V=scripts/ubench/variants
if [ "$1" = build ]; then
  mkdir -p $V
  hipcc --offload-arch=gfx950 -O3 -o $V/ubench_counters scripts/ubench/ubench_counters.hip && ls -la $V/ubench_counters
else
  OUT=${2:-gpurun_out/counters}
  REPO=$PWD
  case "$OUT" in /*) ;; *) OUT=$REPO/$OUT ;; esac
  mkdir -p "$OUT"
  export TMPDIR=/tmp
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt -- $REPO/$V/ubench_counters > "$OUT/true_bytes.json" 2> "$OUT/kt.log"; echo "kt exit $?"
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -o fetch -- $REPO/$V/ubench_counters > /dev/null 2> "$OUT/fetch.log"; echo "fetch exit $?"
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -o write -- $REPO/$V/ubench_counters > /dev/null 2> "$OUT/write.log"; echo "write exit $?"
  cd "$REPO"
  python scripts/counter_calibration.py "$OUT/true_bytes.json" "$(find "$OUT/fetch" -name '*_results.db' | head -1)" \
    "$(find "$OUT/write" -name '*_results.db' | head -1)" "$(find "$OUT/kt" -name '*_results.db' | head -1)" \
    "$OUT/counter_calibration.txt" "$OUT/counter_calibration.json"
  find "$OUT" -name '*.db' -delete
fi
