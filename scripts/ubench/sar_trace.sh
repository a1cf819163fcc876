#!/bin/bash
# Timeline of the small accept-reply kernels (k_ar_small, k_ar_runs<SMALL>): a build with wall-clock stamps per
# workgroup (-DGPX_SAR_TRACE, never shipped), run on a few small calls, summarised by sar_trace_summary.py.
#   build (here):  bash scripts/ubench/sar_trace.sh build
#   run (GPU box): bash scripts/ubench/sar_trace.sh run
cd "$(dirname "$0")/../.."
V=scripts/ubench/variants
if [ "$1" = build ]; then
  mkdir -p $V
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -w -DGPX_SAR_TRACE -o $V/libgpx_SARTRACE.so gigapaxos_amd/csrc/gpx_engine.hip
  ls -la $V/libgpx_SARTRACE.so
else
  GPX_HIP_LIB=$PWD/$V/libgpx_SARTRACE.so timeout 300 python scripts/ubench/sar_trace_run.py
fi
