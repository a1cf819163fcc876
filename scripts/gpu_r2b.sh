#!/bin/bash
# round-2 visit B: front-end microbenchmark, parity at bucket shift 8 and 9, k_hist sub-tile sweep
OUT=gpurun_out/r2b
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 scripts/ubench/ubench_front16.bin > $OUT/ubench_front16.txt 2>&1
echo "ubench exit $?"; cat $OUT/ubench_front16.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
GPX_BUCKET_SHIFT=9 timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_s9.log 2>&1
echo "pytest s9 exit $?" >> $OUT/pytest_gpu_s9.log; tail -3 $OUT/pytest_gpu_s9.log
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline"
run() { tag=$1; shift; env "$@" timeout 300 $B $EXTRA > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python scripts/bench_line.py $tag < $OUT/bench_$tag.json 2>/dev/null || tail -2 $OUT/bench_$tag.err; }
for s in 8 9; do for h in 2 3 5 8; do run s${s}h${h} GPX_BUCKET_SHIFT=$s GPX_HSUB=$h; done; done
EXTRA="--k 5"; run k5s8 GPX_BUCKET_SHIFT=8; run k5s9 GPX_BUCKET_SHIFT=9; run k5legacy GPX_AR_LEGACY=1
EXTRA="--mix"; run mixs8 GPX_BUCKET_SHIFT=8; run mixs9 GPX_BUCKET_SHIFT=9
EXTRA="--sorted"; run sorteds9 GPX_BUCKET_SHIFT=9
EXTRA="--serial"; run serials9 GPX_BUCKET_SHIFT=9
