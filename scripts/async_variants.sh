#!/bin/bash
# bench_async_path.py under the engine's experiment switches (host-pointer path)
for env in "" "GPX_ASYNC_DIRECT=0" "GPX_ASYNC_COPYIN=kernel" "GPX_ASYNC_FILL=memset"; do
  env $env timeout 300 python scripts/bench_async_path.py 2>/dev/null | tail -1
done
