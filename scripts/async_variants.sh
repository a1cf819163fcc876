#!/bin/bash
# bench_async_path.py under the engine's experiment switches (docs/HISTORY.md, host-pointer path)
timeout 300 python scripts/bench_async_path.py --common-first 2>/dev/null | tail -1
for env in "" "GPX_ASYNC_DIRECT=0" "GPX_ASYNC_COPYIN=kernel"; do
  env $env timeout 300 python scripts/bench_async_path.py 2>/dev/null | tail -1
done
