#!/bin/bash
# round 4, visit h: k_ar_small without the prefetch, gather loads in flight together; where it beats the pipeline
OUT=gpurun_out/r04
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_small_ar_gpu.py tests/test_edges_gpu.py -m gpu -q --maxfail=6 > $OUT/tests_h1.log 2>&1
echo "tests h1 exit $?"; tail -3 $OUT/tests_h1.log
bash scripts/ubench/sar_trace.sh run 2>&1 | tee $OUT/sar_trace_4.txt
for g in 1000000 20000; do
for mx in 131072 0; do
echo "== groups $g GPX_SAR_MAX_N=$mx"
GPX_SAR_MAX_N=$mx timeout 300 python scripts/bench_batch_sweep.py --groups $g --min-log2 10 --max-log2 17 2>&1 | tail -1 > $OUT/batch_sweep_h_${g}_$mx.json; python - <<PY
import json
d=json.load(open("$OUT/batch_sweep_h_${g}_$mx.json"))
print({k: v["us_per_call"] for k, v in d["sweep"].items()})
PY
done
done
for mode in "" "--shuffled-replies"; do
  timeout 200 python scripts/bench_full_round.py --groups 10000 --rounds 101 $mode 2>&1 | tail -1 > "$OUT/config2_h$mode.json"
  cut -c1-900 "$OUT/config2_h$mode.json"
done
