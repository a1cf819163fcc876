#!/bin/bash
# round 4, visit l: k_ac_one<.., XCHG> - ordered ACCEPT / COMMIT calls of at most 65,536 records under lazy outputs in one launch
OUT=gpurun_out/r04
mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python -m pytest "tests/test_one_gpu.py::test_lazy_outputs_on_the_device_path" "tests/test_one_gpu.py::test_broken_promise_refuses_from_the_first_violation" \
    tests/test_parity_gpu.py "tests/test_acc_enum_gpu.py::test_acceptor_side_enumerated_under_the_ordered_promise" tests/test_host_cluster_gpu.py \
    -m gpu -q --maxfail=6 --durations=4 -k "ordered or promise or lazy or cluster or config2" > $OUT/tests_m1.log 2>&1
echo "tests l1 exit $?"; tail -9 $OUT/tests_m1.log
for mode in "" "--shuffled-replies"; do
  timeout 200 python scripts/bench_full_round.py --groups 10000 --rounds 101 $mode 2>&1 | tail -1 > "$OUT/config2_m$mode.json"
  cut -c1-900 "$OUT/config2_m$mode.json"
done
timeout 300 python -m pytest tests -m gpu_fast -q -x 2>&1 | tail -2
