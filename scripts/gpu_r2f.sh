#!/bin/bash
OUT=gpurun_out/r2f
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
timeout 300 python scripts/bench_wire.py > $OUT/bench_wire.json 2> $OUT/bench_wire.err; echo "wire $?"; cat $OUT/bench_wire.json; tail -2 $OUT/bench_wire.err
GPX_HIP_LIB=$PWD/gigapaxos_amd/csrc/libgpx_hip_ww1.so timeout 300 python scripts/bench_wire.py > $OUT/bench_wire_ww1.json 2> $OUT/bench_wire_ww1.err; echo "wire ww1 $?"; cat $OUT/bench_wire_ww1.json
timeout 300 python scripts/bench_batch_sweep.py > $OUT/batch_sweep.json 2> $OUT/batch_sweep.err; echo "sweep $?"; python -c "
import json;d=json.load(open('$OUT/batch_sweep.json'));print({k:(v['us_per_call'],round(v['votes_per_sec']/1e9,2)) for k,v in d['sweep'].items()})"
timeout 300 python scripts/bench_full_round.py --groups 10000 --rounds 101 > $OUT/config2.json 2> $OUT/config2.err; echo "config2 $?"; cat $OUT/config2.json
timeout 300 python scripts/small_call_latency.py > $OUT/small_call.json 2>/dev/null; cat $OUT/small_call.json
timeout 300 python scripts/bench_route.py > $OUT/bench_route.json 2> $OUT/bench_route.err; echo "route $?"; cat $OUT/bench_route.json; tail -2 $OUT/bench_route.err
timeout 300 python bench.py --split-global --k 5 --no-cpu-baseline --no-end-to-end > $OUT/bench_split1.json 2> $OUT/bench_split1.err; echo "split $?"; python scripts/bench_line.py split_k5 < $OUT/bench_split1.json
