#!/bin/bash
# First GPU visit after round 3: what was written when no GPU minutes were left.
#   gpurun --timeout 1500 -- 'bash scripts/run_pending_visit.sh'
# 1) the staged GPU legs (tests/test_pending_gpu.py) on the shipped kernels;
# 2) the wire suite and the decode bench with the in-flight staging loop (GPX_WD_STAGE1=1, gpx_wire.hip.h wire_stage:
#    profiles/r03_wire_stage_isa.txt) against the shipped one, twice each.
# Output under gpurun_out/pending/.
OUT=gpurun_out/pending
mkdir -p $OUT
export TMPDIR=/tmp
GPX_RUN_PENDING=1 timeout 900 python -m pytest tests/test_pending_gpu.py -m gpu -q --durations=8 >$OUT/pending_tests.log 2>&1
echo "pending tests exit $?" | tee -a $OUT/pending_tests.log
tail -5 $OUT/pending_tests.log
GPX_WD_STAGE1=1 timeout 600 python -m pytest tests/test_wire_gpu.py -m gpu -q >$OUT/wire_tests_stage1.log 2>&1
echo "wire suite with GPX_WD_STAGE1=1 exit $?" | tee -a $OUT/wire_tests_stage1.log
tail -3 $OUT/wire_tests_stage1.log
for rep in 1 2; do
  for s1 in 0 1; do
    echo "== GPX_WD_STAGE1=$s1 (run $rep)" | tee -a $OUT/bench_wire_stage1.txt
    GPX_WD_STAGE1=$s1 timeout 300 python scripts/bench_wire.py 2>&1 | tail -3 | cut -c1-400 | tee -a $OUT/bench_wire_stage1.txt
  done
done
