#!/bin/bash
# decisions written in place by the per-bucket kernel (look-back over the buckets): parity, then the bench both ways
OUT=gpurun_out/r2r
mkdir -p $OUT
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q > $OUT/pytest_ar.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_ar.log; tail -6 $OUT/pytest_ar.log | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --no-end-to-end > $OUT/bench.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['ms_per_step'], d['value'], d['roofline']['kernels_ms_per_step'], d['roofline']['pipeline_frac'])"
GPX_AR_EMIT_LEGACY=1 timeout 300 python bench.py --no-cpu-baseline --no-end-to-end > $OUT/bench_staged.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/bench_staged.json')); print(d['ms_per_step'], d['value'], d['roofline']['kernels_ms_per_step'], d['roofline']['pipeline_frac'])"
