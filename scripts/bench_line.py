#!/usr/bin/env python
"""Condenses bench.py's JSON line (stdin) to: tag, ms/step, G votes/s, per-kernel microseconds."""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else ""
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1])
    print(tag, d["ms_per_step"], round(d["votes_per_sec"] / 1e9, 2),
          {k: round(v * 1000, 1) for k, v in d["roofline"]["kernels_ms_per_step"].items()})
except Exception as ex:  # noqa: BLE001
    print(tag, "FAILED", ex)
