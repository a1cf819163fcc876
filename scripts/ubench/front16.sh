#!/bin/bash
# scripts/ubench/ubench_front16.hip on the GPU box: bash scripts/ubench/front16.sh [warm]
set -e
cd "$(dirname "$0")"
hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_front16 ubench_front16.hip
/tmp/ubench_front16 "$@"
