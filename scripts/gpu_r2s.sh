#!/bin/bash
# one-launch BATCHED_COMMIT encoder + the full-size full-round parity test
OUT=gpurun_out/r2s
mkdir -p $OUT
timeout 900 python -m pytest tests/test_wire_gpu.py -m gpu -x -q > $OUT/pytest_wire.log 2>&1; echo "pytest wire exit $?" >> $OUT/pytest_wire.log; tail -8 $OUT/pytest_wire.log | cut -c1-300
GPX_PACK_FUSED=0 timeout 300 python scripts/bench_wire.py > $OUT/bench_wire.json 2>$OUT/bench_wire.err; cat $OUT/bench_wire.json
GPX_WIRE_LEGACY=1 timeout 300 python scripts/bench_wire.py > $OUT/bench_wire_legacy.json 2>/dev/null; cat $OUT/bench_wire_legacy.json
timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -x -q -k full_round > $OUT/pytest_full_round.log 2>&1; tail -4 $OUT/pytest_full_round.log | cut -c1-300
