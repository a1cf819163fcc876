#!/bin/bash
OUT=gpurun_out/r2g
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
timeout 300 python scripts/bench_wire.py > $OUT/bench_wire.json 2> $OUT/bench_wire.err; echo "wire $?"; cat $OUT/bench_wire.json; tail -2 $OUT/bench_wire.err
