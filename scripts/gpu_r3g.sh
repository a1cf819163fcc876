#!/bin/bash
OUT=gpurun_out/r3g
mkdir -p $OUT
for args in "--k 5" "--k 5" "--k 5 --mix" ""; do
    timeout 300 python bench.py $args --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$args]', d['ms_per_step'], {k: round(v*1e3,1) for k,v in d['roofline']['kernels_ms_per_step'].items()})"
done
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -k "5 or k5 or kmax5 or steady or mixed_ops" > $OUT/pytest_k5.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_k5.log; tail -4 $OUT/pytest_k5.log | cut -c1-300
