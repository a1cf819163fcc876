#!/bin/bash
# round 4, visit c: the one-launch ordered kernels with checker workgroups
OUT=gpurun_out/r04
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_one_gpu.py tests/test_runs_gpu.py -m gpu -q -x --durations=8 > $OUT/one_tests_c.log 2>&1
echo "one-launch tests exit $?"; tail -12 $OUT/one_tests_c.log
for mode in "" "--dense-always"; do
  timeout 300 python scripts/bench_full_round.py $mode 2>&1 | tail -1 > $OUT/full_round_c$mode.json
  cat $OUT/full_round_c$mode.json | cut -c1-900
done
timeout 300 python bench.py --no-cpu-baseline --no-end-to-end 2>&1 | tail -1 | cut -c1-700 | tee $OUT/bench_quick_c.json
timeout 400 python -m pytest tests -m gpu_fast -q -x --durations=0 > $OUT/gpu_fast_c.log 2>&1
echo "gpu_fast exit $?"; tail -4 $OUT/gpu_fast_c.log
