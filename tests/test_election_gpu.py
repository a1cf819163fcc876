"""Coordinator side of the view change on the GPU: the HIP library against the oracle, bit for bit -
the PaxosCoordinatorState.main restatement, the hand-made cases, and random interleavings of
elections, pre-active proposals, prepare replies and accept replies."""
import pytest

from tests.election_common import (pcs_main_scenario, small_scenarios, boundary_scenario, fuzz_run, V_ELECTED,
                                   V_PREEMPTED)

pytestmark = pytest.mark.gpu


def _same(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert x == y, (x[0], x, y)


def test_pcs_main_parity(hip_lib, oracle_lib):
    _same(pcs_main_scenario(hip_lib), pcs_main_scenario(oracle_lib))


def test_small_scenarios_parity(hip_lib, oracle_lib):
    _same(small_scenarios(hip_lib), small_scenarios(oracle_lib))


def test_half_range_boundary_parity(hip_lib, oracle_lib):
    _same(boundary_scenario(hip_lib), boundary_scenario(oracle_lib))


@pytest.mark.parametrize("seed,G,k,W,steps,slot0", [(1, 96, 3, 8, 60, 0), (2, 300, 5, 16, 60, 0), (3, 700, 3, 8, 80, 0),
                                                    (4, 64, 9, 32, 50, 0), (5, 2000, 3, 8, 40, 0),
                                                    # slots straddling Integer.MAX_VALUE -> MIN_VALUE
                                                    (6, 400, 3, 8, 70, 2 ** 31 - 30), (7, 200, 5, 16, 60, 2 ** 31 - 55)])
def test_election_fuzz_parity(hip_lib, oracle_lib, seed, G, k, W, steps, slot0):
    a = fuzz_run(hip_lib, seed, G=G, k=k, W=W, steps=steps, slot0=slot0)
    b = fuzz_run(oracle_lib, seed, G=G, k=k, W=W, steps=steps, slot0=slot0)
    _same(a, b)
    kinds = [v for x in b if x[0] == "reply" for v in x[1]]
    assert V_ELECTED in kinds and V_PREEMPTED in kinds


@pytest.mark.parametrize("G,seed,window", [(200, 0, 8), (3000, 1, 8), (500, 2, 16)])
def test_failover_end_to_end_parity(hip_lib, oracle_lib, G, seed, window):
    """Three engines: node 0 dies with ACCEPTs in flight, node 1 scans, runs, is elected and carries the
    accepted values over; every step's outputs, the execution logs and the state dumps equal the
    oracle's (the safety invariants are asserted inside failover_run for both)."""
    from tests.failover_common import failover_run

    a = failover_run(hip_lib, G=G, seed=seed, window=window)
    b = failover_run(oracle_lib, G=G, seed=seed, window=window)
    assert a.keys() == b.keys()
    for key in a:
        assert a[key] == b[key], key


def test_election_begin_sequences_against_java_reading(hip_lib):
    """makeCoordinator's begin sequences (PaxosInstanceStateMachine.java:2090-2279) against the Python reading"""
    from tests import test_election_oracle as T
    T.test_election_begin_sequences_against_java_reading(hip_lib)
