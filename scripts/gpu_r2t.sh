#!/bin/bash
# final round-2 visit: all GPU tests, judged line + rocprofv3 trace + traffic counters of the shipped build
OUT=gpurun_out/r2t
mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log | cut -c1-300
bash scripts/gpu_round.sh r02_v3 notest
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
du -sh gpurun_out
