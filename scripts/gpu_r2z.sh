#!/bin/bash
OUT=gpurun_out/r2z
mkdir -p $OUT
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k: round(v*1e3,1) for k,v in d['roofline']['kernels_ms_per_step'].items()})"; done
timeout 1800 python -m pytest tests/test_parity_gpu.py tests/test_fullsize_gpu.py tests/test_configs_gpu.py -m gpu -x -q > $OUT/pytest_ar.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_ar.log; tail -4 $OUT/pytest_ar.log | cut -c1-300
