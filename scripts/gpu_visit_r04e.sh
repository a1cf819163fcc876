#!/bin/bash
# round 4, visit e: k_ar_small (small accept-reply calls in one launch), k_ar_runs<SMALL>, k_pack_one (fused pack)
OUT=gpurun_out/r04
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_small_ar_gpu.py tests/test_edges_gpu.py "tests/test_one_gpu.py::test_lazy_reply_runs" \
    tests/test_runs_gpu.py -m gpu -q --maxfail=6 --durations=8 -k "not 1m_groups or 3-8" > $OUT/tests_e1.log 2>&1
echo "tests e1 exit $?"; tail -25 $OUT/tests_e1.log
timeout 600 python -m pytest tests/test_wire_gpu.py tests/test_parity_gpu.py -m gpu -q --maxfail=6 --durations=8 > $OUT/tests_e2.log 2>&1
echo "tests e2 exit $?"; tail -25 $OUT/tests_e2.log
timeout 300 python scripts/bench_batch_sweep.py 2>&1 | tail -1 > $OUT/batch_sweep_e.json; cut -c1-1800 $OUT/batch_sweep_e.json
GPX_SAR_VOTES_PER_WG=512 timeout 300 python scripts/bench_batch_sweep.py --max-log2 17 2>&1 | tail -1 > $OUT/batch_sweep_e_512.json; cut -c1-600 $OUT/batch_sweep_e_512.json
GPX_SAR_VOTES_PER_WG=2048 timeout 300 python scripts/bench_batch_sweep.py --max-log2 17 2>&1 | tail -1 > $OUT/batch_sweep_e_2048.json; cut -c1-600 $OUT/batch_sweep_e_2048.json
for mode in "" "--shuffled-replies"; do
  timeout 200 python scripts/bench_full_round.py --groups 10000 --rounds 101 $mode 2>&1 | tail -1 > "$OUT/config2_e$mode.json"
  cut -c1-900 "$OUT/config2_e$mode.json"
done
timeout 300 python scripts/bench_wire.py 2>&1 | tail -1 | cut -c1-1000 | tee $OUT/bench_wire_e.json
for a in "--k 5" "--k 5 --runs" "--k 3 --runs"; do
  timeout 200 python bench.py --groups 125000 $a --no-cpu-baseline --no-end-to-end --no-parity-check 2>&1 | tail -1 | cut -c1-700 | tee -a $OUT/shard125k_e.jsonl
done
timeout 300 python bench.py --no-cpu-baseline --no-end-to-end 2>&1 | tail -1 | cut -c1-1500 | tee $OUT/bench_quick_e.json
