#!/bin/bash
# round 4, visit d: check kernel + one work kernel for ordered batches; unregister hazard test; funnel-shift frame writer
OUT=gpurun_out/r04
mkdir -p $OUT
export TMPDIR=/tmp
timeout 700 python -m pytest tests/test_one_gpu.py tests/test_async_gpu.py tests/test_wire_gpu.py -m gpu -q -x --durations=6 > $OUT/tests_d.log 2>&1
echo "tests exit $?"; tail -10 $OUT/tests_d.log
timeout 300 python -m pytest tests -m gpu_fast -q -x > $OUT/gpu_fast_d.log 2>&1
echo "gpu_fast exit $?"; tail -2 $OUT/gpu_fast_d.log
for mode in "" "--dense-always"; do
  timeout 300 python scripts/bench_full_round.py $mode 2>&1 | tail -1 > $OUT/full_round_d$mode.json
  cat $OUT/full_round_d$mode.json | cut -c1-900
done
timeout 300 python bench.py --no-cpu-baseline --no-end-to-end 2>&1 | tail -1 | cut -c1-2200 | tee $OUT/bench_quick_d.json
timeout 300 python scripts/bench_wire.py 2>&1 | tail -1 | cut -c1-900 | tee $OUT/bench_wire_d.json
