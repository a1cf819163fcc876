"""Pins the oracle's restatement of RequestBatcher / roundRobinCoordinator / gap detection
 against hand-computed answers.  CPU only."""
import numpy as np

from gigapaxos_amd import wire as W
from tests import host_rows_common as H


def test_request_batcher_known_answer(oracle_lib):
    H.request_batch_kat(oracle_lib)


def test_request_batcher_invariants(oracle_lib):
    (leader, status, b), _ = H.request_batch_run(oracle_lib, 3)
    leader, b = np.array(leader), {k: np.array(v) for k, v in b.items()}
    assert b["count"].sum() == (np.array(status) == 0).sum()      # every queued request is in a batch
    assert (np.diff(b["gidx"]) >= 0).all()                          # grouped by gidx
    assert (b["bytes"][b["count"] > 1] <= 2000).all() and (b["size"][b["count"] > 1] <= 400).all()
    assert (leader[b["leader"]] == b["leader"]).all()               # heads lead themselves


def test_round_robin_coordinator(oracle_lib):
    """members[Math.abs(ballotnum + paxosID.hashCode()) % k] (PISM:2251-2256), Java int arithmetic."""
    out, names, members, ks = H.coordinator_run(oracle_lib)
    for bi, bal in enumerate((0, 1, 7, -5, 2**31 - 1)):
        for qi, g in enumerate(range(-1, 501)):
            want = -2**31
            if 0 <= g < 480 and g % 9 != 4:
                x = (bal + W.java_string_hash(names[g]) + 2**31) % 2**32 - 2**31
                ax = -x if x < 0 else x
                ax = ax if ax < 2**31 else -2**31
                if ax >= 0:
                    want = int(members[g, ax % int(ks[g])])
            assert out[bi][qi] == want, (bal, g)
    assert out[5] == [-2**31]


def test_gap_detection_known_answer(oracle_lib):
    from gigapaxos_amd import Engine, hri_create, C_HASVALUE
    e = Engine(oracle_lib, 100, 4, kmax=3, window=16)
    we = W.WireEngine(e)
    mem = np.tile(np.array([100, 101, 102], np.int32), (3, 1))
    e.create_groups(np.arange(3), mem, 3, hri_create(3, 3, 100))
    z = lambda n: np.zeros(n, np.int32)  # noqa: E731
    c = lambda n: np.full(n, 100, np.int32)  # noqa: E731
    # group 0: accept for slot 3; decisions 2 (value), 3 (meta, accept present), 5 (meta, no accept), 7 (value)
    e.accept([0], z(1), c(1), [3], z(1))
    e.commit([0, 0, 0, 0], z(4), c(4), [2, 3, 5, 7], z(4), np.array([C_HASVALUE, 0, 0, C_HASVALUE], np.uint8))
    # group 1: executes 1..2, nothing pending
    e.commit([1, 1], z(2), c(2), [1, 2], z(2), np.array([C_HASVALUE, C_HASVALUE], np.uint8))
    first, maxc, missing, sync, st = W.gap_scan(we, [0, 1, 2, 3], threshold=5)
    assert first.tolist() == [1, 3, 1, 0] and maxc.tolist() == [7, 2, 0, 0] and st.tolist() == [0, 0, 0, 1]
    # slots 1..6 examined (i < maxCommittedSlot): 1 missing, 2 ok, 3 ok (meta + accept), 4, 5 (meta, no accept), 6
    assert missing.tolist() == [0b111001, 0, 0, 0]
    assert sync.tolist() == [1, 0, 0, 0]          # gap 6 >= 5; group 1: gap -1
    first, maxc, missing, sync, st = W.gap_scan(we, [0, 1, 2], threshold=100, size_limit=3)
    assert missing.tolist() == [0b001, 0, 0] and sync.tolist() == [1, 0, 0]  # expectedSlot 1 and gap >= 100/100
    assert W.gap_scan(we, [1], 100, W.SYNC_FORCE)[3].tolist() == [1]
    e.close()
