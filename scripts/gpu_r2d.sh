#!/bin/bash
OUT=gpurun_out/r2d
mkdir -p $OUT
bash scripts/gpu_round.sh r02_v1
timeout 300 python scripts/bench_full_round.py > $OUT/full_round.json 2> $OUT/full_round.err; echo "full_round $?"; cat $OUT/full_round.json; tail -2 $OUT/full_round.err
timeout 300 python scripts/bench_full_round.py --no-promise > $OUT/full_round_nopromise.json 2> $OUT/full_round_np.err; echo "full_round np $?"; cat $OUT/full_round_nopromise.json
timeout 300 python bench.py --no-cpu-baseline --no-promise > $OUT/bench_nopromise.json 2>/dev/null; python scripts/bench_line.py nopromise < $OUT/bench_nopromise.json
