#!/bin/bash
# round-2 profile visit: judged line + rocprofv3 trace + traffic counters, SQ / LDS counters of the
# bench and of the wire path, side benches of the final build
OUT=gpurun_out/r2l
mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -40 $OUT/pytest_gpu.log | cut -c1-300
bash scripts/gpu_round.sh r02_v2 notest
bash scripts/gpu_pmc.sh r02_pmc > $OUT/pmc_bench.log 2>&1; tail -5 $OUT/pmc_bench.log
GPX_PMC_CMD="python $PWD/scripts/bench_wire.py --rounds 3" bash scripts/gpu_pmc.sh r02_pmc_wire > $OUT/pmc_wire.log 2>&1; tail -3 $OUT/pmc_wire.log
export TMPDIR=/tmp
timeout 300 python scripts/bench_batch_sweep.py > $OUT/batch_sweep.json 2>/dev/null; python -c "
import json;d=json.load(open('$OUT/batch_sweep.json'));print({k:(v['us_per_call'],round(v['votes_per_sec']/1e9,2)) for k,v in d['sweep'].items()})"
timeout 300 python scripts/bench_full_round.py --groups 10000 --rounds 101 > $OUT/config2.json 2>/dev/null; cat $OUT/config2.json
timeout 300 python scripts/small_call_latency.py > $OUT/small_call.json 2>/dev/null; cat $OUT/small_call.json
timeout 300 python scripts/bench_wire.py > $OUT/bench_wire.json 2>/dev/null; cat $OUT/bench_wire.json
timeout 300 python scripts/bench_host_path.py > $OUT/host_path.json 2>/dev/null; cat $OUT/host_path.json
scripts/ubench/ubench_front16.bin > $OUT/ubench.txt 2>&1; tail -12 $OUT/ubench.txt
du -sh gpurun_out
