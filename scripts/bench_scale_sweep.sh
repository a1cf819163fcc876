#!/bin/bash
# bench.py at other sizes and streams (one JSON object per line); numbers go to docs/HISTORY.md §5
#   gpurun --timeout 1200 -- 'bash scripts/bench_scale_sweep.sh > gpurun_out/scale_sweep.jsonl'
run() { timeout 400 python bench.py "$@" --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'args': '$*', 'ms_per_step': d['ms_per_step'], 'decisions_per_sec': d['value'], 'votes_per_sec': d['votes_per_sec'],
  'pipeline_frac': d['roofline']['pipeline_frac'], 'kernels_ms_per_step': d['roofline']['kernels_ms_per_step']}))"; }
for G in 250000 1000000 2000000 4000000 8000000 16000000; do run --groups $G; done
run --k 5
run --mix
run --k 5 --mix
run --sorted
run --no-promise
