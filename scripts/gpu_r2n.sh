#!/bin/bash
# one-launch wire decode + name rows with the (exists, version) copies: parity first, then the numbers
OUT=gpurun_out/r2n
mkdir -p $OUT
timeout 600 python -m pytest tests/test_wire_gpu.py -m gpu -x -q > $OUT/pytest_wire.log 2>&1; echo "pytest wire exit $?" >> $OUT/pytest_wire.log; tail -15 $OUT/pytest_wire.log | cut -c1-300
timeout 300 python scripts/bench_wire.py > $OUT/bench_wire.json 2>$OUT/bench_wire.err; cat $OUT/bench_wire.json; tail -3 $OUT/bench_wire.err
GPX_WIRE_LEGACY=1 timeout 300 python scripts/bench_wire.py > $OUT/bench_wire_legacy.json 2>/dev/null; cat $OUT/bench_wire_legacy.json
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.log | cut -c1-300
du -sh gpurun_out
