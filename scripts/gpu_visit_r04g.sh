#!/bin/bash
# round 4, visit g: k_ar_small third build (no draw, slice check merged, LDS-DMA prefetch), k_ar_runs<SMALL> with the
# verdict exchanged through tickets, pack back to two launches
OUT=gpurun_out/r04
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_small_ar_gpu.py tests/test_edges_gpu.py "tests/test_one_gpu.py::test_lazy_reply_runs" \
    tests/test_runs_gpu.py tests/test_parity_gpu.py -m gpu -q --maxfail=6 --durations=5 -k "not 1m_groups or 3-8" > $OUT/tests_g1.log 2>&1
echo "tests g1 exit $?"; tail -12 $OUT/tests_g1.log
timeout 200 python -m pytest tests/test_wire_gpu.py -m gpu -q -x -k "pack or commit or codec" > $OUT/tests_g2.log 2>&1
echo "tests g2 exit $?"; tail -3 $OUT/tests_g2.log
for pf in 1 0; do
echo "== GPX_SAR_PREFETCH=$pf"
GPX_SAR_PREFETCH=$pf bash scripts/ubench/sar_trace.sh run 2>&1 | tee $OUT/sar_trace_3_pf$pf.txt
done
for mx in 131072 0; do
GPX_SAR_MAX_N=$mx timeout 300 python scripts/bench_batch_sweep.py --min-log2 13 --max-log2 17 2>&1 | tail -1 > $OUT/batch_sweep_g_$mx.json; cut -c1-1200 $OUT/batch_sweep_g_$mx.json
done
for mode in "" "--shuffled-replies"; do
  timeout 200 python scripts/bench_full_round.py --groups 10000 --rounds 101 $mode 2>&1 | tail -1 > "$OUT/config2_g$mode.json"
  cut -c1-900 "$OUT/config2_g$mode.json"
done
timeout 300 python scripts/bench_wire.py 2>&1 | tail -1 | cut -c1-1000 | tee $OUT/bench_wire_g.json
