"""The C++ host layer (gpx::PaxosManager over the C-ABI) on the CPU: its in-process cluster built
against the oracle.  Checks the host logic - frame building, forwarding to the coordinator,
loopback short circuits, value bookkeeping, in-order upcalls: every replica executes every request
exactly once, slot == sequence number (TESTPaxosApp's invariant), identical hash chains."""
import pytest

from tests.host_cluster_common import CASES, build_oracle_cluster, run_cluster


@pytest.mark.parametrize("case", range(len(CASES)))
def test_cluster_invariants(case):
    out = run_cluster(build_oracle_cluster(), CASES[case])
    assert out["ok"] is True
    assert out["executed_per_node"] == out["requests"] == out["groups"] * out["rounds"]
    for n in out["per_node"]:
        assert n["executed"] == out["requests"] and n["dropped_frames"] == 0 and n["refused"] == 0
    assert sum(n["proposed"] for n in out["per_node"]) == out["requests"]
    assert sum(n["decisions"] for n in out["per_node"]) == out["requests"]
    if "--entry" not in CASES[case]:
        assert sum(n["forwarded"] for n in out["per_node"]) > 0 or out["groups"] == 1
