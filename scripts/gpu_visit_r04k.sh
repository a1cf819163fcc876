#!/bin/bash
# round 4, visit k: k_ac_small / k_propose_small with the group state requested ahead of the verdict, straight-line scan
OUT=gpurun_out/r04
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_edges_gpu.py tests/test_host_rows_gpu.py tests/test_host_cluster_gpu.py \
    tests/test_async_gpu.py "tests/test_one_gpu.py::test_lazy_outputs_on_the_device_path" "tests/test_one_gpu.py::test_broken_promise_refuses_from_the_first_violation" \
    "tests/test_acc_enum_gpu.py::test_acceptor_side_enumerated_under_the_ordered_promise" \
    "tests/test_acc_enum_gpu.py::test_whole_round_against_the_two_java_readings_together_on_engine" \
    "tests/test_acc_enum_gpu.py::test_acceptor_side_at_the_int_wrap" -m gpu -q --maxfail=6 --durations=5 > $OUT/tests_k1.log 2>&1
echo "tests k1 exit $?"; tail -12 $OUT/tests_k1.log
for mode in "" "--shuffled-replies"; do
  timeout 200 python scripts/bench_full_round.py --groups 10000 --rounds 101 $mode 2>&1 | tail -1 > "$OUT/config2_k$mode.json"
  cut -c1-900 "$OUT/config2_k$mode.json"
done
timeout 100 python scripts/small_call_latency.py 2>&1 | tail -1 | tee $OUT/small_call_latency_k.json
