#!/bin/bash
# Timeline of k_wire_decode1: a build with wall-clock stamps per tile (-DGPX_WD_TRACE, never shipped),
# run on bench_wire.py's burst, summarised by wire_trace_summary.py.
#   build (here):  bash scripts/ubench/wire_trace.sh build
#   run (GPU box): bash scripts/ubench/wire_trace.sh run     (CALL = which decode call of bench_wire.py: 0-3 the 2 M
#                  reply frames, 4-6 the 1 M ACCEPT frames with 64-byte values; TILES = tile sizes)
cd "$(dirname "$0")/../.."
V=scripts/ubench/variants
if [ "$1" = build ]; then
  mkdir -p $V
  for v in "" NOLOOKBACK NOLOOKUP; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -w -DGPX_WD_TRACE ${v:+-DGPX_WD_$v} -o $V/libgpx_TRACE$(echo "$v" | sed "s/ -DGPX_//g; s/=//g").so gigapaxos_amd/csrc/gpx_engine.hip &
  done
  wait
  ls -la $V/libgpx_TRACE*.so
else
  for v in "" ${VARIANTS}; do
  for t in ${TILES:-256 512}; do
    echo "== GPX_WD_TILE=$t $v"
    GPX_BENCH_NOCHECK=1 GPX_WD_TILE=$t GPX_WD_TRACE_CALL=${CALL:-2} GPX_WD_TRACE_FILE=/tmp/wd_trace_$t.bin GPX_HIP_LIB=$PWD/$V/libgpx_TRACE$v.so timeout 300 python scripts/bench_wire.py --rounds 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('decode_ms', d['decode_ms'], 'k_wire_decode1', d['kernels_us'].get('k_wire_decode1'))"
    python scripts/ubench/wire_trace_summary.py /tmp/wd_trace_$t.bin
  done
  done
fi
