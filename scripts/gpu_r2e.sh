#!/bin/bash
OUT=gpurun_out/r2e
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
timeout 300 python scripts/small_call_latency.py > $OUT/small_call.json 2>$OUT/small_call.err; echo "small $?"; cat $OUT/small_call.json
GPX_SMALL=0 timeout 300 python scripts/small_call_latency.py > $OUT/small_call_nosmall.json 2>/dev/null; cat $OUT/small_call_nosmall.json
timeout 300 python scripts/bench_batch_sweep.py > $OUT/batch_sweep.json 2> $OUT/batch_sweep.err; echo "sweep $?"; cat $OUT/batch_sweep.json
timeout 300 python scripts/bench_full_round.py --groups 10000 --rounds 101 > $OUT/config2.json 2> $OUT/config2.err; echo "config2 $?"; cat $OUT/config2.json
GPX_SMALL=0 timeout 300 python scripts/bench_full_round.py --groups 10000 --rounds 101 > $OUT/config2_nosmall.json 2>/dev/null; cat $OUT/config2_nosmall.json
timeout 300 python scripts/bench_full_round.py --groups 125000 --rounds 21 > $OUT/shard125k.json 2>/dev/null; cat $OUT/shard125k.json
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench $?"; python scripts/bench_line.py default < $OUT/bench.json; python -c "
import json;d=json.load(open('$OUT/bench.json'));print(d['end_to_end']);print(d['parity_checked'])"
timeout 300 python bench.py --groups 125000 --no-cpu-baseline --no-end-to-end > $OUT/bench_125k.json 2>/dev/null; python scripts/bench_line.py 125k < $OUT/bench_125k.json
timeout 300 python bench.py --groups 125000 --k 5 --no-cpu-baseline --no-end-to-end > $OUT/bench_125k_k5.json 2>/dev/null; python scripts/bench_line.py 125k_k5 < $OUT/bench_125k_k5.json
