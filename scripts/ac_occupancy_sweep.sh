#!/bin/bash
# k_ac_direct at a forced occupancy (amdgpu_waves_per_eu 6 / 7 / 8; the product build takes what the register
# allocator gives: 89 / 80 VGPRs = 5 / 6 waves per SIMD): the full round with each library
#   build (here):  bash scripts/ac_occupancy_sweep.sh build      run (GPU box): bash scripts/ac_occupancy_sweep.sh
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p scripts/ubench/variants
  for w in 6 7 8; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -w -DGPX_AC_WAVES=$w -o scripts/ubench/variants/libgpx_ACW$w.so gigapaxos_amd/csrc/gpx_engine.hip &
  done
  wait
  ls -la scripts/ubench/variants/libgpx_ACW*.so
  exit 0
fi
for v in "" ACW6 ACW7 ACW8; do
  echo "== ${v:-product}"
  for r in 1 2; do
    ${v:+env GPX_HIP_LIB=$PWD/scripts/ubench/variants/libgpx_$v.so} timeout 300 python scripts/bench_full_round.py 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_round'], d['phases_ms'], 'k_ac_direct', d['kernels_us_per_round']['k_ac_direct'])"
  done
done
