#!/bin/bash
# does k_bucket_ar16's time step with ceil(workgroups / resident slots)?  768 slots = 256 CUs x 3 workgroups
for G in 393216 400000 589824 600000 786432 800000 1000000 1179648 1200000; do timeout 300 python bench.py --groups $G --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']; print($G, ($G+511)//512, round(($G+511)//512/768,2), d['ms_per_step'], {a: round(v*1e3,1) for a,v in k.items()})"; done
