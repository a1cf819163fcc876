#!/bin/bash
# round 4, visit f: k_ar_small second build (mask scan, one thread per group), k_ar_runs<SMALL> straight-line scan
OUT=gpurun_out/r04
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_small_ar_gpu.py tests/test_edges_gpu.py "tests/test_one_gpu.py::test_lazy_reply_runs" \
    tests/test_runs_gpu.py tests/test_parity_gpu.py -m gpu -q --maxfail=6 --durations=5 -k "not 1m_groups or 3-8" > $OUT/tests_f1.log 2>&1
echo "tests f1 exit $?"; tail -12 $OUT/tests_f1.log
bash scripts/ubench/sar_trace.sh run 2>&1 | tee $OUT/sar_trace_2.txt
timeout 300 python scripts/bench_batch_sweep.py --max-log2 18 2>&1 | tail -1 > $OUT/batch_sweep_f.json; cut -c1-900 $OUT/batch_sweep_f.json
for v in 512 256; do
GPX_SAR_VOTES_PER_WG=$v timeout 300 python scripts/bench_batch_sweep.py --max-log2 17 2>&1 | tail -1 > $OUT/batch_sweep_f_$v.json; cut -c1-600 $OUT/batch_sweep_f_$v.json
done
for mode in "" "--shuffled-replies"; do
  timeout 200 python scripts/bench_full_round.py --groups 10000 --rounds 101 $mode 2>&1 | tail -1 > "$OUT/config2_f$mode.json"
  cut -c1-900 "$OUT/config2_f$mode.json"
done
