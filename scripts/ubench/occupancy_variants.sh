#!/bin/bash
# k_bucket_ar16 at other register budgets (builds under scripts/ubench/variants):
#   W8   amdgpu_waves_per_eu(8): 64 VGPRs + 60 B of scratch      W7  waves_per_eu(7)
#   NP   coordinator state fetched after the regrouping          W8NP both
# build (here, no GPU needed), one library per variant under scripts/ubench/variants/ (git-ignored):
#   cd gigapaxos_amd/csrc && for v in "W8:-DGPX_AR16_WAVES=8" "W7:-DGPX_AR16_WAVES=7" "NP:-DGPX_B16_NOPRELOAD" \
#       "W8NP:-DGPX_AR16_WAVES=8 -DGPX_B16_NOPRELOAD"; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -w \
#       ${v#*:} -o ../../scripts/ubench/variants/libgpx_${v%%:*}.so gpx_engine.hip; done
cd "$(dirname "$0")/../.."
for f in "" scripts/ubench/variants/libgpx_*.so; do
  echo "== ${f:-product build}"
  if [ -n "$f" ]; then export GPX_HIP_LIB=$PWD/$f; else unset GPX_HIP_LIB; fi
  timeout 200 python bench.py --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k: round(v*1e3,1) for k,v in d['roofline']['kernels_ms_per_step'].items()}, d.get('parity_checked'))"
done
