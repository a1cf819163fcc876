#!/bin/bash
# Where does k_wire_decode1's time go?  Builds of the library with one stage of the kernel removed
# (compile-time switches in gpx_wire.hip.h, results wrong on purpose), timed with bench_wire.py.
#   build (here):  bash scripts/ubench/wire_ablation.sh build
#   run (GPU box): bash scripts/ubench/wire_ablation.sh run > gpurun_out/wire_ablation.txt
cd "$(dirname "$0")/../.."
V=scripts/ubench/variants
if [ "$1" = build ]; then
  mkdir -p $V
  for v in NOLOOKUP NOEMIT NOPARSE NOLOOKBACK NOSTAGE "NOPARSE -DGPX_WD_NOEMIT" "NOPARSE -DGPX_WD_NOEMIT -DGPX_WD_NOSTAGE"; do
    name=$(echo "$v" | sed 's/ -DGPX_WD_/_/g')
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -w -DGPX_WD_$v -o $V/libgpx_$name.so gigapaxos_amd/csrc/gpx_engine.hip &
  done
  wait
  ls -la $V
else
  for f in $V/libgpx_*.so; do
    echo "== $f"
    GPX_BENCH_NOCHECK=1 GPX_HIP_LIB=$PWD/$f timeout 200 python scripts/bench_wire.py --rounds 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('decode_ms', d['decode_ms'], 'accept_decode_ms', d['accept_decode_ms'], 'k_wire_decode1', d['kernels_us'].get('k_wire_decode1'))"
  done
fi
