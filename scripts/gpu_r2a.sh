#!/bin/bash
# round-2 visit A: parity of the 16-byte vote record back end + A/B sweeps of its knobs
OUT=gpurun_out/r2a
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/nproc.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline"
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout 300 $B > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  echo "$tag exit $?"; python scripts/bench_line.py $tag < $OUT/bench_$tag.json 2>/dev/null || tail -2 $OUT/bench_$tag.err
}
run s8 GPX_BUCKET_SHIFT=8
run s9 GPX_BUCKET_SHIFT=9
run s10 GPX_BUCKET_SHIFT=10
run legacy GPX_AR_LEGACY=1
run s8w8 GPX_BUCKET_SHIFT=8 GPX_HIP_LIB=$PWD/gigapaxos_amd/csrc/libgpx_hip_w8.so
run s9w8 GPX_BUCKET_SHIFT=9 GPX_HIP_LIB=$PWD/gigapaxos_amd/csrc/libgpx_hip_w8.so
env GPX_BUCKET_SHIFT=9 timeout 300 $B --k 5 > $OUT/bench_k5s9.json 2> $OUT/bench_k5s9.err; python scripts/bench_line.py k5s9 < $OUT/bench_k5s9.json
env timeout 300 $B --k 5 > $OUT/bench_k5s8.json 2> $OUT/bench_k5s8.err; python scripts/bench_line.py k5s8 < $OUT/bench_k5s8.json
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?"; cat $OUT/bench.json; tail -3 $OUT/bench.err
