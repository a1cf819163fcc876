#!/bin/bash
OUT=gpurun_out/r2h
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
B="python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-end-to-end"
run() { tag=$1; shift; env "$@" timeout 300 $B $EXTRA > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  python scripts/bench_line.py $tag < $OUT/bench_$tag.json 2>/dev/null || tail -2 $OUT/bench_$tag.err; }
EXTRA=""; run serial X=1
EXTRA="--pipelined"; run pipe_light X=1; run pipe_full GPX_PIPE_FULL=1
EXTRA=""; run w7 GPX_HIP_LIB=$PWD/gigapaxos_amd/csrc/libgpx_hip_w7.so
EXTRA="--k 5"; run k5 X=1
EXTRA="--k 5 --pipelined"; run k5_pipe X=1
for v in "" "--unordered"; do
  timeout 300 python scripts/bench_full_round.py $v > $OUT/full_round$v.json 2> $OUT/full_round$v.err; echo "full_round $v $?"; cat $OUT/full_round$v.json
done
GPX_AR_LEGACY=1 timeout 300 python scripts/bench_full_round.py --unordered > $OUT/full_round_unordered_legacy.json 2>/dev/null; cat $OUT/full_round_unordered_legacy.json
