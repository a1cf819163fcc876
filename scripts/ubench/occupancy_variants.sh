#!/bin/bash
# k_bucket_ar16 at other register budgets (builds under scripts/ubench/variants):
#   W8   amdgpu_waves_per_eu(8): 64 VGPRs + 60 B of scratch      W7  waves_per_eu(7)
#   NP   coordinator state fetched after the regrouping          W8NP both
cd "$(dirname "$0")/../.."
for f in "" scripts/ubench/variants/libgpx_*.so; do
  echo "== ${f:-product build}"
  GPX_HIP_LIB=${f:+$PWD/$f} timeout 200 python bench.py --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k: round(v*1e3,1) for k,v in d['roofline']['kernels_ms_per_step'].items()}, d.get('parity_checked'))"
done
