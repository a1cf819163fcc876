#!/bin/bash
for args in "" "" "" "--k 5" "--k 5"; do
    timeout 300 python bench.py $args --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('[$args]', d['ms_per_step'], {k: round(v*1e3,1) for k,v in d['roofline']['kernels_ms_per_step'].items()})"
done
