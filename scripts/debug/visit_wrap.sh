#!/bin/bash
# GPU visit: where does the engine leave the oracle at the int wrap (tee), then the staged wrap legs again
OUT=gpurun_out/wrap
mkdir -p $OUT
export TMPDIR=/tmp
for c in 0 1; do
  timeout 300 python scripts/debug/tee_engines.py wrap $c > $OUT/tee_case$c.log 2>&1
  echo "tee case $c exit $?"; tail -3 $OUT/tee_case$c.log | cut -c1-300
done
GPX_RUN_PENDING=1 timeout 600 python -m pytest tests/test_pending_gpu.py -m gpu -q -k "int_wrap or half_the_int" > $OUT/wrap_tests.log 2>&1
echo "wrap tests exit $?"; tail -6 $OUT/wrap_tests.log
timeout 300 python -m pytest tests/test_wire_gpu.py tests/test_async_gpu.py -m gpu -q -x > $OUT/wire_async.log 2>&1
echo "wire+async exit $?"; tail -3 $OUT/wire_async.log
