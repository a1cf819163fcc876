#!/bin/bash
OUT=gpurun_out/r2i
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end"
for g in 250000 1000000 2000000 4000000 8000000 16000000; do
  timeout 600 $B --groups $g > $OUT/scale_g$g.json 2> $OUT/scale_g$g.err; python scripts/bench_line.py g$g < $OUT/scale_g$g.json || tail -3 $OUT/scale_g$g.err
done
timeout 600 $B --groups 1000000 --k 5 > $OUT/scale_k5.json 2>/dev/null; python scripts/bench_line.py k5 < $OUT/scale_k5.json
timeout 600 $B --groups 1000000 --mix > $OUT/scale_mix.json 2>/dev/null; python scripts/bench_line.py mix < $OUT/scale_mix.json
timeout 600 $B --groups 1000000 --sorted > $OUT/scale_sorted.json 2>/dev/null; python scripts/bench_line.py sorted < $OUT/scale_sorted.json
