#!/bin/bash
for lib in "" scripts/ubench/variants/libgpx_prev.so; do
  for args in "--mix" "--k 5 --mix" "--k 5"; do
    if [ -n "$lib" ]; then export GPX_HIP_LIB=$PWD/$lib; else unset GPX_HIP_LIB; fi
    timeout 300 python bench.py $args --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('${lib:-product}', '$args', d['ms_per_step'], {k: round(v*1e3,1) for k,v in d['roofline']['kernels_ms_per_step'].items()})"
  done
done
