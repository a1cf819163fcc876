#!/bin/bash
# k_wire_decode1 with 256 / 512 frames per workgroup (GPX_WD_TILE): parity first, then bench_wire.py
cd "$(dirname "$0")/.."
for t in 256 512; do
  echo "== GPX_WD_TILE=$t"
  GPX_WD_TILE=$t timeout 600 python -m pytest tests/test_wire_gpu.py -m gpu -q -x  2>&1 | tail -2
  for r in 1 2; do
    GPX_WD_TILE=$t timeout 200 python scripts/bench_wire.py --rounds 6 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('decode_ms', d['decode_ms'], 'accept_decode_ms', d['accept_decode_ms'], 'k_wire_decode1', d['kernels_us'].get('k_wire_decode1'), 'pack_ms', d['pack_commits_ms'])"
  done
done
