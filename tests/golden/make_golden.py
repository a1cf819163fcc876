#!/usr/bin/env python
"""Regenerates tests/golden/*.npz.

NOT reference-generated: the reference is Java and no JVM exists in the build image or on the GPU
box, so these vectors come from the CPU ORACLE (oracle/gpx_oracle.cpp, oracle/gpx_wire_oracle.inc).
They freeze the oracle's answers on seeded scenarios so that neither the oracle nor the engine can
drift silently: tests/test_golden.py replays the scenarios on the oracle (CPU) and on the HIP engine
(GPU) and compares with the files bit for bit.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from tests.golden_scenarios import SCENARIOS  # noqa: E402
from tests.oracle_binding import load_oracle  # noqa: E402


def main():
    lib = load_oracle()
    for name, fn in SCENARIOS.items():
        out = fn(lib)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, {k: v.shape for k, v in out.items()}, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
