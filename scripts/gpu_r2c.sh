#!/bin/bash
# round-2 visit C: parity with the direct paths in, judged bench line, full-round and side benches
OUT=gpurun_out/r2c
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?"; python scripts/bench_line.py default < $OUT/bench.json; tail -2 $OUT/bench.err
timeout 300 python bench.py --no-cpu-baseline --pipelined > $OUT/bench_pipe.json 2> $OUT/bench_pipe.err; python scripts/bench_line.py pipelined < $OUT/bench_pipe.json
timeout 300 python scripts/bench_full_round.py > $OUT/full_round.json 2> $OUT/full_round.err; echo "full_round $?"; cat $OUT/full_round.json; tail -2 $OUT/full_round.err
timeout 300 python scripts/bench_wire.py > $OUT/bench_wire.json 2> $OUT/bench_wire.err; echo "wire $?"; cat $OUT/bench_wire.json | head -30
timeout 300 python scripts/bench_batch_sweep.py > $OUT/batch_sweep.json 2> $OUT/batch_sweep.err; echo "sweep $?"; tail -12 $OUT/batch_sweep.json
timeout 300 python scripts/small_call_latency.py > $OUT/small_call.txt 2>&1; echo "small $?"; tail -12 $OUT/small_call.txt
timeout 300 python scripts/bench_host_path.py > $OUT/host_path.json 2> $OUT/host_path.err; echo "host $?"; cat $OUT/host_path.json
