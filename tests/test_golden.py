"""Replays the frozen scenarios of tests/golden/ (oracle-generated vectors: see make_golden.py —
the Java reference cannot be run here) on the oracle (CPU) and on the HIP engine (GPU)."""
import os

import numpy as np
import pytest

from tests.golden_scenarios import SCENARIOS

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _check(name, lib):
    want = np.load(os.path.join(GOLDEN, name + ".npz"))
    got = SCENARIOS[name](lib)
    assert sorted(want.files) == sorted(got)
    for k in want.files:
        assert want[k].shape == got[k].shape and (want[k] == got[k]).all(), f"{name}.{k}"


@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_oracle_reproduces_golden(oracle_lib, name):
    _check(name, oracle_lib)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_engine_reproduces_golden(hip_lib, name):
    _check(name, hip_lib)
