"""The C++ host layer on the GPU: gigapaxos_amd/host/gpx_loopback_cluster (linked against
libgpx_hip.so) must print exactly what the same program built against the oracle prints - state
digest, per-node counters, frame and byte counts."""
import pytest

from tests.host_cluster_common import CASES_GPU as CASES, build_hip_cluster, build_oracle_cluster, run_cluster

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", range(len(CASES)))
def test_cluster_matches_oracle_build(hip_lib, case):
    a = run_cluster(build_hip_cluster(), CASES[case])
    b = run_cluster(build_oracle_cluster(), CASES[case])
    assert a == b
    assert a["ok"] is True
