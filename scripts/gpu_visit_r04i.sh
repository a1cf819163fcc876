#!/bin/bash
# round 4, visit i: k_ar_tiny (one workgroup, at most 1,024 votes)
OUT=gpurun_out/r04
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_small_ar_gpu.py tests/test_edges_gpu.py tests/test_parity_gpu.py -m gpu -q --maxfail=6 > $OUT/tests_j1.log 2>&1
echo "tests i1 exit $?"; tail -5 $OUT/tests_j1.log
bash scripts/ubench/sar_trace.sh run 2>&1 | tee $OUT/sar_trace_6.txt
for g in 1000000 20000; do
for mx in 1024 0; do
echo "== groups $g GPX_SAR_MAX_N=$mx"
GPX_SAR_MAX_N=$mx timeout 300 python scripts/bench_batch_sweep.py --groups $g --min-log2 7 --max-log2 11 2>&1 | tail -1 > $OUT/batch_sweep_j_${g}_$mx.json; python - <<PY
import json
d=json.load(open("$OUT/batch_sweep_j_${g}_$mx.json"))
print({k: v["us_per_call"] for k, v in d["sweep"].items()})
PY
done
done
timeout 100 python scripts/small_call_latency.py 2>&1 | tail -1 | tee $OUT/small_call_latency_j.json
