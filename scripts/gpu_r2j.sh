#!/bin/bash
OUT=gpurun_out/r2j
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
timeout 300 python scripts/bench_full_round.py --unordered > $OUT/full_round_unordered.json 2> $OUT/fr.err; cat $OUT/full_round_unordered.json; tail -2 $OUT/fr.err
timeout 300 python scripts/bench_full_round.py > $OUT/full_round.json 2>/dev/null; cat $OUT/full_round.json
