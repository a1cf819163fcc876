#!/bin/bash
# three bench lines (kernel times) of the current build; optional: GPX_HIP_LIB
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k: round(v*1e3,1) for k,v in d['roofline']['kernels_ms_per_step'].items()})"; done
