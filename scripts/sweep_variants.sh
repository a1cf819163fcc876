#!/bin/bash
# bench every library build under scratch/variants/ (tuning aid; see scripts/gpu_round.sh for the judged run)
for f in scratch/variants/*.so; do
  echo "== $f $*"
  GPX_HIP_LIB=$PWD/$f python bench.py --no-cpu-baseline --steps 10 "$@" 2>&1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], {k:round(v*1000,1) for k,v in d['roofline']['kernels_ms_per_step'].items()})"
done
