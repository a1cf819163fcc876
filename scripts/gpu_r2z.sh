#!/bin/bash
# DPP wave scans + lean runtime block scan: bench, then every GPU test
OUT=gpurun_out/r2z
mkdir -p $OUT
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k: round(v*1e3,1) for k,v in d['roofline']['kernels_ms_per_step'].items()})"; done
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log | cut -c1-300
timeout 300 python scripts/bench_full_round.py > $OUT/full_round.json 2>/dev/null; cat $OUT/full_round.json
timeout 300 python scripts/bench_wire.py > $OUT/bench_wire.json 2>/dev/null; cat $OUT/bench_wire.json | cut -c1-700
