#!/bin/bash
# round 4, visit b: the one-launch ordered kernels (parity, then the full round's time) and the slotted-scatter ubench
OUT=gpurun_out/r04
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_one_gpu.py tests/test_parity_gpu.py tests/test_runs_gpu.py tests/test_edges_gpu.py -m gpu -q -x --durations=10 > $OUT/one_tests.log 2>&1
echo "one-launch tests exit $?"; tail -15 $OUT/one_tests.log
for mode in "" "--dense-always"; do
  timeout 300 python scripts/bench_full_round.py $mode 2>&1 | tail -1 > $OUT/full_round$mode.json
  cat $OUT/full_round$mode.json | cut -c1-900
done
timeout 300 python bench.py --no-cpu-baseline --no-end-to-end 2>&1 | tail -1 | cut -c1-1500 | tee $OUT/bench_quick.json
(cd scripts/ubench && hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -Wno-unused-result -o /tmp/ub16 ubench_front16.hip 2>&1 | grep -E "error" -A3; timeout 300 /tmp/ub16 > ../../$OUT/ubench_front16_D.txt 2>&1)
grep -E "^D|overflow|k_read_region|^A shift  9 nbk 1954 hsub 3|64-byte|128-byte" $OUT/ubench_front16_D.txt
