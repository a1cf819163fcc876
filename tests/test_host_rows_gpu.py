"""RequestBatcher / roundRobinCoordinator / gap detection on the HIP engine vs the oracle
(identical scenarios, every output compared exactly)."""
import pytest

from tests import host_rows_common as H

pytestmark = pytest.mark.gpu


def test_request_batcher_known_answer(hip_lib):
    H.request_batch_kat(hip_lib)


@pytest.mark.parametrize("seed,hot", [(1, True), (2, False)])
def test_request_batcher_matches_oracle(hip_lib, oracle_lib, seed, hot):
    a = H.request_batch_run(hip_lib, seed, hot=hot)
    b = H.request_batch_run(oracle_lib, seed, hot=hot)
    for (la, sa, ba), (lb, sb, bb) in zip(a, b):
        assert sa == sb and la == lb
        assert ba == bb


def test_round_robin_coordinator_matches_oracle(hip_lib, oracle_lib):
    assert H.coordinator_run(hip_lib)[0] == H.coordinator_run(oracle_lib)[0]


@pytest.mark.parametrize("seed", [0, 1])
def test_gap_detection_matches_oracle(hip_lib, oracle_lib, seed):
    assert H.gap_run(hip_lib, seed) == H.gap_run(oracle_lib, seed)


def test_election_scan_known_answer(hip_lib):
    from tests.test_host_rows_oracle import test_election_scan_known_answer as kat
    kat(hip_lib)


@pytest.mark.parametrize("seed", [0, 1])
def test_election_scan_matches_oracle(hip_lib, oracle_lib, seed):
    assert H.election_run(hip_lib, seed) == H.election_run(oracle_lib, seed)


def test_request_batcher_and_election_scan_against_java_reading(hip_lib):
    """the Python readings of RequestBatcher.dequeueImpl (RequestBatcher.java:182-239) and checkRunForCoordinator
    (PaxosInstanceStateMachine.java:2090-2279) on the engine"""
    from tests import test_host_rows_oracle as T
    T.test_request_batcher_random_bursts_against_java_reading(hip_lib)
    T.test_election_scan_random_groups_against_java_reading(hip_lib)
