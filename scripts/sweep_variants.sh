#!/bin/bash
# bench every library build under scratch/variants/ (tuning aid; see scripts/gpu_visit.sh for the judged run)
#   sweep_variants.sh [bench.py flags]      env: SWEEP_ENV="A=1 B=2" extra environment per run; SWEEP_REPS=n passes (interleaved)
for rep in $(seq 1 ${SWEEP_REPS:-1}); do
for f in scratch/variants/*.so; do
  echo "== $f $* $SWEEP_ENV"
  env $SWEEP_ENV GPX_HIP_LIB=$PWD/$f python bench.py --no-cpu-baseline --no-end-to-end --steps 20 "$@" 2>&1 | python -c "
import sys,json
txt=sys.stdin.read().strip().splitlines()
try:
    d=json.loads(txt[-1])
    print(d['ms_per_step'], {k:round(v*1000,1) for k,v in d['roofline']['kernels_ms_per_step'].items()})
except Exception as e:
    print('FAILED', txt[-3:])"
done
done
