#!/bin/bash
# Timeline of the tiled accept-reply call's two kernels (k_scatter_tiles, k_bucket_ar16_tiles): a build with wall-clock
# stamps per workgroup (-DGPX_TL_TRACE, never shipped), run on BASELINE's shapes, summarised per phase.
#   build (here):  bash scripts/ubench/tiles_trace.sh build
#   run (GPU box): bash scripts/ubench/tiles_trace.sh run [G K]...
cd "$(dirname "$0")/../.."
V=scripts/ubench/variants
if [ "$1" = build ]; then
  mkdir -p $V
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -w -DGPX_TL_TRACE -o $V/libgpx_TLTRACE.so gigapaxos_amd/csrc/gpx_engine.hip
  ls -la $V/libgpx_TLTRACE.so
else
  shift
  GPX_HIP_LIB=$PWD/$V/libgpx_TLTRACE.so timeout 300 python scripts/ubench/tiles_trace_run.py "$@"
fi
