#!/bin/bash
# parity tests of the single-launch acceptor kernels after the status fix + partition microbenchmark C
OUT=gpurun_out/r2m
mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -30 $OUT/pytest_gpu.log | cut -c1-300
scripts/ubench/ubench_front16.bin > $OUT/ubench.txt 2>&1; cat $OUT/ubench.txt
timeout 300 python scripts/bench_full_round.py --groups 10000 --rounds 101 > $OUT/config2.json 2>/dev/null; cat $OUT/config2.json
du -sh gpurun_out
