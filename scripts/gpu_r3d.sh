#!/bin/bash
# side benches + SQ / TCC counters of the final round-2 build
OUT=gpurun_out/r3d
mkdir -p $OUT
export TMPDIR=/tmp
bash scripts/gpu_pmc.sh r02_pmc_final > $OUT/pmc_bench.log 2>&1; tail -3 $OUT/pmc_bench.log
timeout 300 python scripts/bench_batch_sweep.py > $OUT/batch_sweep.json 2>/dev/null; python -c "
import json;d=json.load(open('$OUT/batch_sweep.json'));print({k:(v['us_per_call'],round(v['votes_per_sec']/1e9,2)) for k,v in d['sweep'].items()})"
timeout 300 python scripts/bench_full_round.py --groups 10000 --rounds 101 > $OUT/config2.json 2>/dev/null; cat $OUT/config2.json
timeout 300 python scripts/bench_full_round.py > $OUT/full_round.json 2>/dev/null; cat $OUT/full_round.json
timeout 300 python scripts/bench_full_round.py --unordered > $OUT/full_round_unordered.json 2>/dev/null; cat $OUT/full_round_unordered.json
timeout 300 python scripts/small_call_latency.py > $OUT/small_call.json 2>/dev/null; cat $OUT/small_call.json
timeout 300 python scripts/bench_host_path.py > $OUT/host_path.json 2>/dev/null; cat $OUT/host_path.json
