"""Pins the CPU oracle against every known-answer / property assertion the reference's own
self-tests hold for this path.  The reference has no golden-vector files and
cannot be run here (no JVM), so these restated assertions are the pinning there is.

Each test names the reference self-test it restates.
"""
import ctypes as C

import numpy as np
import pytest

from gigapaxos_amd import (Engine, hri_create, hri_initial, make_hri, S_OK, S_FORWARD, S_REFUSED,
                           S_STOPPED, S_NOGROUP, S_EXISTS, S_BUSY, D_DECISION, D_PREEMPTED, R_TOLOG,
                           R_STORED, A_STOP, C_HASVALUE, RETIRE_PAUSE, RETIRE_KILL)
from gigapaxos_amd.loopback import LoopbackCluster


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_waitfor_main_known_answer(oracle_lib):
    """WaitforUtility.main (paxosutil/WaitforUtility.java:145-161): members {0,9,4,23};
    9 -> changed, no majority; 23 -> changed, no majority; 0 -> changed, majority."""
    members = np.array([0, 9, 4, 23], np.int32)
    nodes = np.array([9, 23, 0], np.int32)
    out = np.zeros(3, np.uint8)
    heard = oracle_lib.lib.orc_waitfor_trace(_p(members), 4, _p(nodes), 3, _p(out))
    assert out.tolist() == [2, 2, 3]
    assert heard == 3
    # !contains(32): a non-member is silently ignored; duplicates are idempotent
    nodes = np.array([32, 9, 9, 4], np.int32)
    out = np.zeros(4, np.uint8)
    heard = oracle_lib.lib.orc_waitfor_trace(_p(members), 4, _p(nodes), 4, _p(out))
    assert out.tolist() == [0, 2, 0, 2] and heard == 2


def test_ballot_compare_wraparound(oracle_lib):
    """Ballot.compareTo (paxosutil/Ballot.java:60-73) incl. the int-subtraction wraparound."""
    f = oracle_lib.lib.orc_ballot_compare
    assert f(1, 5, 1, 5) == 0
    assert f(2, 0, 1, 9) == 1
    assert f(1, 4, 1, 9) == -1
    # wraparound: MIN_VALUE is "after" MAX_VALUE
    assert f(-2147483648, 0, 2147483647, 0) == 1
    assert f(2147483647, 0, -2147483648, 0) == -1
    # PaxosPacketBatcher.main (PaxosPacketBatcher.java:556-567): Ballot(23,456) equals itself
    assert f(23, 456, 23, 456) == 0


def test_median_minus(oracle_lib):
    """PaxosCoordinatorState.getMedianMinus (PaxosCoordinatorState.java:867-875)."""
    f = oracle_lib.lib.orc_median_minus

    def med(xs):
        a = np.array(xs, np.int32)
        return f(_p(a), len(xs))

    assert med([5]) == 5
    assert med([3, 1, 2]) == 2  # odd: index k/2
    assert med([4, 1, 3, 2]) == 2  # even: index k/2-1
    assert med([0, 0, 0]) == 0
    assert med([-1, 7, 7, -1, 3]) == 3
    rng = np.random.default_rng(1)
    for k in range(1, 17):
        xs = rng.integers(-50, 50, k).astype(np.int32)
        idx = k // 2 - 1 if k % 2 == 0 else k // 2
        assert med(xs.tolist()) == np.sort(xs)[idx]


def _java_string_hash(s: str) -> int:
    h = 0
    for ch in s.encode("latin-1"):
        h = (31 * h + ch) & 0xFFFFFFFF
    return h - (1 << 32) if h >= (1 << 31) else h


def test_round_robin_coordinator(oracle_lib):
    """PISM.roundRobinCoordinator (PaxosInstanceStateMachine.java:2251-2256) with Java
    String.hashCode: "hello".hashCode() == 99162322 is the well-known Java value."""
    assert _java_string_hash("hello") == 99162322
    f = oracle_lib.lib.orc_round_robin_coordinator
    members = np.array([100, 101, 102], np.int32)
    assert f(b"hello", _p(members), 3, 0) == members[99162322 % 3]
    for name in ["paxos0", "TESTPaxosApp17", "a", "zzzzzzzzzzzzzzzz", "service_name_42"]:
        for b in (0, 1, 7):
            x = (_java_string_hash(name) + b + (1 << 31)) % (1 << 32) - (1 << 31)
            ax = -x if x < 0 else x
            if ax >= (1 << 31):  # Math.abs(MIN_VALUE)
                continue
            assert f(name.encode(), _p(members), 3, b) == members[ax % 3]


def _mk(oracle_lib, my_id, members, n_groups=1, window=64, rows=None):
    k = len(members)
    e = Engine(oracle_lib, my_id, n_groups, kmax=k, window=window)
    mem = np.tile(np.array(members, np.int32), (n_groups, 1))
    if rows is None:
        rows = hri_create(n_groups, k, my_id)
    st = e.create_groups(np.arange(n_groups), mem, k, rows)
    assert (st == S_OK).all()
    return e


def test_pcs_main_accept_reply_section(oracle_lib):
    """PaxosCoordinatorState.main, "Testing accept replies" (PaxosCoordinatorState.java:
    1166-1212): the reference uses 43 members; the ABI caps groups at PC.MAX_GROUP_SIZE = 16
    (PaxosConfig.java:532), so 15 here.  For each outstanding proposal feed a reply from every
    member, a few % of them with a higher ballot.  Assertions restated: a returned pvalue is a DECISION (my ballot)
    or PREEMPTED (higher ballot); the slot is gone from myProposals afterwards; DECISION only
    after a majority; myProposals ends empty (or the coordinator resigned)."""
    rng = np.random.default_rng(7)
    members = [21]
    for _ in range(14):
        members.append(members[-1] + 1 + int(rng.integers(0, 10)))
    rows = make_hri(1)
    rows["acc_bnum"] = 2
    rows["acc_bcoord"] = 21
    rows["has_coord"] = 1
    rows["coord_bnum"] = 2
    rows["coord_bcoord"] = 21
    rows["node_slots"][:] = -1
    e = _mk(oracle_lib, 21, members, rows=rows)
    nprop = 9
    slot, bnum, bcoord, median, st = e.propose(np.zeros(nprop, np.int32))
    assert (st == S_OK).all() and slot.tolist() == list(range(nprop))
    assert (bnum == 2).all() and (bcoord == 21).all() and (median == -1).all()
    resigned = False
    for s in range(nprop):
        heard = 0
        done = False
        for m in members:
            higher = rng.random() > 0.96
            d = e.accept_reply([0], [3 if higher else 2], [21], [s], [m], [-1])
            if resigned:
                assert d.gidx.shape[0] == 0
                continue
            if higher:
                if not done:
                    assert d.kind.tolist() == [D_PREEMPTED]
                    done = True
                else:
                    assert d.gidx.shape[0] == 0
            else:
                heard += 1
                if not done and heard > len(members) // 2:
                    assert d.kind.tolist() == [D_DECISION] and d.slot.tolist() == [s]
                    assert d.bnum.tolist() == [2] and d.bcoord.tolist() == [21]
                    done = True
                else:
                    assert d.gidx.shape[0] == 0
            dump = e.dump(0)
            has_coord = _parse_dump(dump)["coord"] is not None
            if not has_coord:
                resigned = True
    st = _parse_dump(e.dump(0))
    assert st["coord"] is None or st["coord"]["proposals"] == []


def _parse_dump(w):
    w = list(map(int, w))
    i = 0
    out = {}
    exists = w[i]; i += 1
    if not exists:
        return None
    out["version"] = w[i]; i += 1
    k = w[i]; i += 1
    out["members"] = w[i:i + k]; i += k
    out["acc"] = dict(slot=w[i], bnum=w[i + 1], bcoord=w[i + 2], gc=w[i + 3], stopped=w[i + 4]); i += 5
    na = w[i]; i += 1
    out["accepted"] = [tuple(w[i + 4 * j:i + 4 * j + 4]) for j in range(na)]; i += 4 * na
    nc = w[i]; i += 1
    out["committed"] = [tuple(w[i + 6 * j:i + 6 * j + 6]) for j in range(nc)]; i += 6 * nc
    hc = w[i]; i += 1
    out["coord"] = None
    if hc:
        c = dict(bnum=w[i], bcoord=w[i + 1], next=w[i + 2]); i += 3
        c["node_slots"] = w[i:i + k]; i += k
        npn = w[i]; i += 1
        c["proposals"] = [tuple(w[i + 3 * j:i + 3 * j + 3]) for j in range(npn)]; i += 3 * npn
        c["active"] = w[i]; i += 1
        if not c["active"]:  # running for coordinator: waitfor, pre-active handles, carry-overs
            c["waitfor"] = (w[i], w[i + 1]); i += 2
            i += 2 * npn
            nco = w[i]; i += 1
            c["carryover"] = [tuple(w[i + 6 * j:i + 6 * j + 6]) for j in range(nco)]; i += 6 * nco
        out["coord"] = c
    assert i == len(w)
    return out


def test_acceptor_monotone_ballot_property(oracle_lib):
    """PaxosAcceptor.testAcceptor (PaxosAcceptor.java:749-776): acceptor(22:1, slot 7);
    100k random accepts: response ballot >= request ballot and >= previous ballot."""
    rows = make_hri(1)
    rows["acc_slot"] = 7
    rows["acc_bnum"] = 22
    rows["acc_bcoord"] = 1
    rows["acc_gc_slot"] = -1
    # window = 0: the ORACLE without the engine's ring limits - the reference's unbounded maps, which is what
    # testAcceptor runs on (slots all over the int range would collide in any ring)
    e = _mk(oracle_lib, 9, [9, 10, 11], window=0, rows=rows)
    rng = np.random.default_rng(3)
    n = 100000
    bnum = rng.integers(0, 2**31 - 1, n).astype(np.int32)
    bcoord = rng.integers(0, 2**31 - 1, n).astype(np.int32)
    slot = rng.integers(0, 2**31 - 1, n).astype(np.int32)
    (rb, rc, rmax, rfl, st), runs = e.accept(np.zeros(n, np.int32), bnum, bcoord, slot,
                                             np.full(n, -1, np.int32))
    assert (st == S_OK).all()
    f = oracle_lib.lib.orc_ballot_compare
    prev = (22, 1)
    for i in range(0, n, 97):  # sample (python loop)
        assert f(int(rb[i]), int(rc[i]), int(bnum[i]), int(bcoord[i])) >= 0
    # monotone over the whole stream
    r64 = rb.astype(np.int64) * (1 << 32) + rc.astype(np.int64)
    assert (np.diff(r64) >= 0).all()
    assert r64[0] >= 22 * (1 << 32) + 1
    assert runs.gidx.shape[0] == 0


def test_hri_roundtrip(oracle_lib):
    """HotRestoreInfoTest.testToStringAndBack (HotRestoreInfo.java:159-175) restated on the
    binary row: create from a row, pause, get the identical row back.
    members {1,4,67}, accSlot 5, accBallot 3:4, gc 3, coordBallot 45:67, next 34, nodeSlots {1,3,5}."""
    rows = make_hri(1)
    rows["version"] = 2
    rows["acc_slot"] = 5
    rows["acc_bnum"] = 3
    rows["acc_bcoord"] = 4
    rows["acc_gc_slot"] = 3
    rows["has_coord"] = 1
    rows["coord_bnum"] = 45
    rows["coord_bcoord"] = 67
    rows["next_proposal_slot"] = 34
    rows["node_slots"][0, :3] = [1, 3, 5]
    e = _mk(oracle_lib, 67, [1, 4, 67], rows=rows)
    snap, st = e.snapshot([0])
    assert st.tolist() == [S_OK]
    assert snap.tobytes() == rows.tobytes()
    back, st = e.retire_groups([0], RETIRE_PAUSE)
    assert st.tolist() == [S_OK]
    assert back.tobytes() == rows.tobytes()
    # gone now
    _, st = e.snapshot([0])
    assert st.tolist() == [S_NOGROUP]
    # a node that is not the coordinator restores no coordinator (PISM:682-684)
    e2 = _mk(oracle_lib, 4, [1, 4, 67], rows=rows)
    snap, _ = e2.snapshot([0])
    assert snap["has_coord"][0] == 0 and snap["next_proposal_slot"][0] == -1


def test_accept_tolog_and_gc_rules(oracle_lib):
    """handleAccept's toLog rule (PaxosInstanceStateMachine.java:1146-1149) and
    acceptAndUpdateBallot / garbageCollectAccepted (PaxosAcceptor.java:302-322, 476-494)."""
    e = _mk(oracle_lib, 101, [100, 101, 102], rows=hri_create(1, 3, 100))
    z = np.zeros(1, np.int32)
    # fresh accept at current ballot: stored + logged; reply maxcp = slot-1 = 0
    (rb, rc, rmax, rfl, st), _ = e.accept(z, [0], [100], [1], [0])
    assert (rb[0], rc[0], rmax[0], rfl[0]) == (0, 100, 0, R_TOLOG | R_STORED)
    # same accept again: stored (put) but NOT logged (prev.ballot == ballot)
    (_, _, _, rfl, _), _ = e.accept(z, [0], [100], [1], [0])
    assert rfl[0] == R_STORED
    # lower ballot: NACK with current ballot, nothing stored/logged
    (rb, rc, _, rfl, _), _ = e.accept(z, [0], [99], [2], [0])
    assert (rb[0], rc[0], rfl[0]) == (0, 100, 0)
    # higher ballot: adopted
    (rb, rc, _, rfl, _), _ = e.accept(z, [1], [102], [1], [0])
    assert (rb[0], rc[0], rfl[0]) == (1, 102, R_TOLOG | R_STORED)
    d = _parse_dump(e.dump(0))
    assert d["acc"] == dict(slot=1, bnum=1, bcoord=102, gc=0, stopped=0)
    assert d["accepted"] == [(1, 1, 102, 0)]
    # slot <= gcSlot: not stored, not logged
    (_, _, _, rfl, _), _ = e.accept(z, [1], [102], [0], [0])
    assert rfl[0] == 0


def test_batched_commit_placeholder_then_accept(oracle_lib):
    """a BATCHED_COMMIT slot without a matching stored ACCEPT queues a value-less
    placeholder; execution stalls until the ACCEPT arrives (PISM:1492, 1158-1161;
    PaxosAcceptor.java:369-385)."""
    e = _mk(oracle_lib, 101, [100, 101, 102], rows=hri_create(1, 3, 100))
    z = np.zeros(1, np.int32)
    st, runs = e.commit(z, [0], [100], [1], [0])
    assert st.tolist() == [S_OK] and runs.gidx.shape[0] == 0
    d = _parse_dump(e.dump(0))
    assert d["committed"] == [(1, 0, 100, 0, 0, 0)]
    # commit for slot 2 with a stored accept: still stalls behind slot 1
    e.accept(z, [0], [100], [2], [0])
    st, runs = e.commit(z, [0], [100], [2], [0])
    assert runs.gidx.shape[0] == 0
    # the accept for slot 1 arrives: releases 1 and 2 in one run
    (_, _, rmax, _, _), runs = e.accept(z, [0], [100], [1], [0])
    assert rmax.tolist() == [0]  # reply is built BEFORE the release
    assert runs.as_tuple_array().tolist() == [[0, 1, 2]]
    d = _parse_dump(e.dump(0))
    assert d["acc"]["slot"] == 3 and d["committed"] == [] and d["accepted"] == []
    # mismatching ballot: placeholder, not executed
    e.accept(z, [0], [100], [3], [0])
    st, runs = e.commit(z, [1], [102], [3], [0])
    assert runs.gidx.shape[0] == 0
    assert _parse_dump(e.dump(0))["committed"] == [(3, 1, 102, 0, 0, 0)]
    # a full DECISION (value at host) for the same slot overrides the placeholder
    st, runs = e.commit(z, [1], [102], [3], [0], [C_HASVALUE])
    assert runs.as_tuple_array().tolist() == [[0, 3, 1]]


def test_stop_request_semantics(oracle_lib):
    """Stop: no proposal after a stop (PaxosCoordinatorState.java:235-239); executing a stop
    stops the acceptor and clears committedRequests (PaxosAcceptor.java:462-474); later packets
    are dropped (PaxosInstanceStateMachine.java:456-460)."""
    c = LoopbackCluster(oracle_lib, [100, 101, 102], 2, window=16)
    c.round([0, 1])
    c.round([0, 1], is_stop=[1, 0])
    slot, _, _, _, st = c.engines[100].propose([0, 1])
    assert st.tolist() == [S_STOPPED, S_OK]
    for nid in (100, 101, 102):
        d = _parse_dump(c.engines[nid].dump(0))
        assert d["acc"]["stopped"] == 1 and d["acc"]["slot"] == 3
        st, _ = c.engines[nid].commit([0], [0], [100], [5], [0])
        assert st.tolist() == [S_STOPPED]
    # refused: stop proposed but not yet decided
    e = _mk(oracle_lib, 100, [100, 101, 102])
    e.propose([0], [1])
    _, _, _, _, st = e.propose([0])
    assert st.tolist() == [S_REFUSED]


def test_preemption_and_resignation(oracle_lib):
    """higher-ballot reply removes just that slot; the coordinator object is dropped
    only when myProposals is then empty; lower-ballot replies are ignored; late votes still
    update nodeSlotNumbers (§9.6)."""
    e = _mk(oracle_lib, 100, [100, 101, 102])
    e.propose([0, 0])  # slots 1, 2
    d = e.accept_reply([0], [1], [101], [1], [101], [0])
    assert d.as_tuple_array().tolist() == [[0, 1, 0, 100, -1, D_PREEMPTED]]
    assert _parse_dump(e.dump(0))["coord"] is not None
    # lower ballot ignored with no side effect
    before = e.dump(0).tolist()
    d = e.accept_reply([0], [0], [99], [2], [101], [5])
    assert d.gidx.shape[0] == 0 and e.dump(0).tolist() == before
    # two votes decide slot 2; third (late) vote only bumps nodeSlots
    d = e.accept_reply([0, 0, 0], [0] * 3, [100] * 3, [2] * 3, [100, 101, 102], [1, 1, 7])
    assert d.as_tuple_array().tolist() == [[0, 2, 0, 100, 1, D_DECISION]]
    st = _parse_dump(e.dump(0))
    assert st["coord"]["node_slots"] == [1, 1, 7] and st["coord"]["proposals"] == []
    # higher ballot with nothing outstanding: coordinator resigns even though nothing preempted
    d = e.accept_reply([0], [1], [101], [9], [101], [0])
    assert d.gidx.shape[0] == 0
    assert _parse_dump(e.dump(0))["coord"] is None
    # afterwards proposals are forwarded to the acceptor's ballot coordinator
    _, bnum, bcoord, _, st = e.propose([0])
    assert st.tolist() == [S_FORWARD] and bcoord.tolist() == [100]


def test_config1_loopback_one_group(oracle_lib):
    """BASELINE config #1 semantics (tests/loopback_1_group: nodes 100..102, 1 group,
    NUM_REQUESTS=10000): every replica executes slots 1..10000 in order, identical streams —
    the TESTPaxosApp invariant `state.seqnum == requestPacket.slot` (TESTPaxosApp.java:190)."""
    c = LoopbackCluster(oracle_lib, [100, 101, 102], 1, window=8)
    n = 10000
    for r in range(n):
        dec = c.round([0])
        assert dec.tolist() == [[0, r + 1, 0, 100, max(r - 0, 0) if r == 0 else dec[0, 4], D_DECISION]]
    ref = None
    for nid in (100, 101, 102):
        ex = c.executed(nid)
        slots = np.concatenate([np.arange(f, f + cnt) for _, f, cnt in ex])
        assert slots.tolist() == list(range(1, n + 1))
        ref = slots if ref is None else ref
        assert (slots == ref).all()
    # medians are monotone and trail the slot
    med = np.concatenate(c.decision_log)[:, 4]
    assert (np.diff(med) >= 0).all() and med[-1] == n - 1


def test_lifecycle_statuses(oracle_lib):
    e = Engine(oracle_lib, 100, 4, kmax=3, window=8)
    mem = np.tile(np.array([100, 101, 102], np.int32), (2, 1))
    assert e.create_groups([0, 7], mem, 3, hri_create(2, 3, 100)).tolist() == [S_OK, S_NOGROUP]
    assert e.create_groups([0], mem[:1], 3, hri_create(1, 3, 100)).tolist() == [S_EXISTS]
    e.propose([0])
    _, st = e.retire_groups([0, 1], RETIRE_PAUSE)
    assert st.tolist() == [S_BUSY, S_NOGROUP]
    _, st = e.retire_groups([0], RETIRE_KILL)
    assert st.tolist() == [S_OK]
    d = e.accept_reply([0, -1, 9], [0] * 3, [100] * 3, [1] * 3, [100] * 3, [0] * 3)
    assert d.status.tolist() == [S_NOGROUP] * 3


def test_prepare_acceptor_side(oracle_lib):
    """PaxosAcceptor.handlePrepare (PaxosAcceptor.java:239-273): adopt a strictly higher ballot, NACK
    (no pvalues) a lower one, return accepted pvalues with slot >= firstUndecidedSlot, gcSlot =
    max(acceptedGCSlot, firstUndecidedSlot - 1); log the prepare iff the ballot went up
    (PISM:985-993); a stopped acceptor does not answer."""
    from gigapaxos_amd import Engine, hri_create, S_OK, S_STOPPED, S_NOGROUP, A_STOP, C_HASVALUE, C_STOP
    e = Engine(oracle_lib, 101, 4, kmax=3, window=8)
    mem = np.tile(np.array([100, 101, 102], np.int32), (3, 1))
    assert (e.create_groups(np.arange(3), mem, 3, hri_create(3, 3, 100)) == S_OK).all()
    z = lambda n: np.zeros(n, np.int32)  # noqa: E731
    c = lambda n: np.full(n, 100, np.int32)  # noqa: E731
    e.accept([0, 0, 0], z(3), c(3), [1, 2, 3], z(3))            # group 0 accepted slots 1..3 at (0,100)
    (rb, rc, rg, rf, st), rows = e.prepare([0, 0, 0, 1, 3], [0, 1, 0, 5, 0], [100, 102, 100, 101, 100],
                                           [2, 2, 1, 1, 1])
    assert st.tolist() == [S_OK, S_OK, S_OK, S_OK, S_NOGROUP]
    assert list(zip(rb.tolist(), rc.tolist()))[:4] == [(0, 100), (1, 102), (1, 102), (5, 101)]
    assert rf.tolist() == [0, 2, 1, 2, 0]                        # ack; upgrade -> TOLOG; NACK; upgrade
    assert rg.tolist()[:4] == [1, 1, 0, 0]                       # max(gc = -1, first - 1)
    assert rows == [(0, 2, 0, 100), (0, 3, 0, 100), (1, 2, 0, 100), (1, 3, 0, 100)]  # none for the NACK
    # a stopped acceptor (executed a stop decision) answers nothing
    e.commit([2], z(1), c(1), [1], z(1), np.array([C_HASVALUE | C_STOP], np.uint8))
    (rb, rc, rg, rf, st), rows = e.prepare([2], [9], [102], [1])
    assert st.tolist() == [S_STOPPED] and rows == [] and rb.tolist() == [0]
    assert e.dump(2).tolist()[:20] == e.dump(2).tolist()[:20]
    e.close()


def test_hri_string_form_roundtrip(oracle_lib):
    """HotRestoreInfoTest.testToStringAndBack (HotRestoreInfo.java:159-175) on the STRING form the
    reference keeps in its pause table: the row an engine hands back on pause, written as
    HotRestoreInfo.toString would (:102-122), parsed as HotRestoreInfo(String) does (:86-98), restores
    the identical group - and the test's own literal."""
    from gigapaxos_amd import hri_to_string, hri_from_string
    rows = make_hri(1)
    rows["version"], rows["acc_slot"], rows["acc_bnum"], rows["acc_bcoord"], rows["acc_gc_slot"] = 2, 5, 3, 4, 3
    rows["has_coord"], rows["coord_bnum"], rows["coord_bcoord"], rows["next_proposal_slot"] = 1, 45, 67, 34
    rows["node_slots"][0, :3] = [1, 3, 5]
    str1 = hri_to_string("paxos0", [1, 4, 67], rows[0])
    assert str1 == "paxos0|2|[1,4,67]|5|3:4|3|45:67|34|[1,3,5]"   # what the Java prints for hri1
    name, members, back = hri_from_string(str1)
    assert name == "paxos0" and members == [1, 4, 67] and back.tobytes() == rows.tobytes()
    assert hri_to_string(name, members, back[0]) == str1
    # through an engine: create from the parsed row, pause, the string of the returned row is the same
    e = _mk(oracle_lib, 67, members, rows=back)
    paused, st = e.retire_groups([0], RETIRE_PAUSE)
    assert st.tolist() == [S_OK] and hri_to_string(name, members, paused[0]) == str1
    # a node without the coordinator: coordBallot and nodeSlots are "null" (:114-121)
    e2 = _mk(oracle_lib, 4, members, rows=back)
    paused, _ = e2.retire_groups([0], RETIRE_PAUSE)
    s2 = hri_to_string(name, members, paused[0])
    assert s2 == "paxos0|2|[1,4,67]|5|3:4|3|null|-1|null"
    assert hri_from_string(s2)[2].tobytes() == paused.tobytes()


@pytest.mark.parametrize("K,nprop", [(3, 2), (4, 2), (5, 2), (3, 3)])
def test_pcs_main_accept_reply_tail_every_coin(oracle_lib, K, nprop):
    """PaxosCoordinatorState.main's accept-reply tail with Math.random enumerated (tests/pcs_enum_common.py):
    2^(K * nprop) reply patterns, the oracle against a statement-by-statement Python reading of the Java."""
    from tests.pcs_enum_common import run_all
    assert run_all(oracle_lib, K, nprop) == 1 << (K * nprop)


def test_acceptor_side_enumerated_against_java_reading(oracle_lib):
    """The acceptor side (handleAccept, handleBatchedCommit, handleCommittedRequest,
    extractExecuteAndCheckpoint, putAndRemoveNextExecutable, reconstructDecision, executed, both garbage
    collectors: PaxosAcceptor.java:302-385, 462-506; PISM:1080-1166, 1432-1528, 1619-1701) as a
    statement-by-statement Python reading of the Java (tests/acc_enum_common.py, written from the
    reference, not from the oracle): every op sequence of length <= 2 over 300 ops, a seeded 30 % of the
    length-3 / length-4 sequences over smaller alphabets (the GPU test runs all of them) and seeded random
    sequences of length 5 and 6 - every reply, status, execution run, final row and (sampled) the maps."""
    import tests.acc_enum_common as A
    for k in A.COVERAGE:
        A.COVERAGE[k] = 0
    # (the whole plan on the oracle: profiles/r03_acc_enum_full_plan_oracle.txt; here the 90,000 pairs see one initial
    # row per batch order instead of both, and the in-memory-accepts pass leaves them to the GPU test)
    n = A.run_plan(oracle_lib, scale=0.1, both_inits_below=50_000)
    n += A.run_plan(oracle_lib, scale=0.03, from_disk=(False,), skip=("len2-wide",))  # GET_ACCEPTED_PVALUES_FROM_DISK = false
    assert n > 400_000
    assert all(v > 0 for v in A.COVERAGE.values()), A.COVERAGE


@pytest.mark.parametrize("K,nprop,G,nv", [(3, 3, 25_000, 24), (5, 4, 12_000, 40), (4, 2, 15_000, 16), (3, 6, 8_000, 60)])
def test_pcs_accept_replies_in_any_order_against_java_reading(oracle_lib, K, nprop, G, nv):
    """The coordinator side beyond PaxosCoordinatorState.main's loop: every group gets its own random stream of
    accept replies - any member, any slot (outstanding, decided long ago, never proposed), duplicates, lower /
    own / higher ballots, checkpoint slots - against tests/pcs_enum_common.model_stream, the same
    statement-by-statement reading of PaxosCoordinator.handleAcceptReply (:210-250) and PCS:597-683, 809-825."""
    from tests.pcs_enum_common import run_streams
    assert run_streams(oracle_lib, K, nprop, G, nv, seed=K * 100 + nprop) == G
    # ... and with 5 % of the votes from a node that is no member of the group (WaitforUtility.getIndex = -1)
    assert run_streams(oracle_lib, K, nprop, G // 2, nv, seed=K * 100 + nprop + 7, p_stranger=0.05) == G // 2
    # ... and checkpoint slots half the int range apart: recordSlotNumber's plain < is not Ballot's wraparound compare
    assert run_streams(oracle_lib, K, nprop, G // 4, nv, seed=K * 100 + nprop + 9, p_extreme=0.1) == G // 4
    # ... and coordinators restored at nextProposalSlot 1 + base: the proposals' slots cross Integer.MAX_VALUE
    for base in (2**31 - 3, 2**31 - 1):
        assert run_streams(oracle_lib, K, nprop, G // 8, nv, seed=K * 100 + nprop + 11, p_extreme=0.05, base=base) == G // 8


@pytest.mark.parametrize("G,rounds,seed,p_drop,K,p_rival", [(3000, 20, 2, 0.2, 3, 0.0), (4000, 16, 3, 0.05, 3, 0.0),
                                                            (1500, 30, 4, 0.35, 3, 0.0), (2000, 20, 5, 0.2, 5, 0.0),
                                                            (1500, 16, 6, 0.1, 4, 0.0), (3000, 20, 21, 0.1, 3, 0.03),
                                                            (2000, 16, 22, 0.2, 5, 0.05), (3000, 24, 41, 0.1, 3, -0.02),
                                                            (2000, 20, 42, 0.15, 5, 0.03)])
def test_whole_round_against_the_two_java_readings_together(oracle_lib, G, rounds, seed, p_drop, K, p_rival):
    """propose -> ACCEPT x 3 -> accept replies -> decision -> BATCHED_COMMIT x 3 -> execution with lost and
    retransmitted messages - and, in the last cases, a rival's ACCEPTs in a higher ballot: NACKs, preempted
    proposals, coordinators that resign, requests forwarded; STOP requests: proposals refused behind an
    outstanding stop, instances that stop and drop everything after - three (four, five) oracle engines against tests/round_model.py (the
    coordinator reading and the acceptor reading of the Java composed; neither written from the oracle)."""
    from tests.round_model import run_rounds
    p_stop = 0.02 if seed > 40 else 0.0   # the last two cases: 2 % of the requests are STOP requests (p_rival < 0: no rival)
    checked, executed = run_rounds(oracle_lib, G, rounds, seed, p_drop=p_drop, K=K, p_rival=max(p_rival, 0.0), p_stop=p_stop,
                                   from_disk=seed % 2 == 0)   # odd seeds: GET_ACCEPTED_PVALUES_FROM_DISK = false
    assert checked > G * rounds * 3 and executed > G * rounds // 5
    assert (run_rounds.resigned > G // 10) == (p_rival > 0.0)  # a rival's higher ballot preempts, and only that
    assert (run_rounds.stopped > G // 4 and run_rounds.refused > 0 and run_rounds.stopped_props > G) == (p_stop > 0.0)


@pytest.mark.parametrize("G,rounds,seed,p_drop,K,p_rival,p_stop", [(3000, 12, 71, 0.1, 3, 0.0, 0.0), (2500, 20, 72, 0.15, 3, 0.0, 0.02),
                                                                   (2000, 16, 73, 0.2, 5, 0.03, 0.0), (2000, 14, 74, 0.3, 4, 0.0, 0.0)])
def test_view_change_after_lossy_rounds_against_java_reading(oracle_lib, G, rounds, seed, p_drop, K, p_rival, p_stop):
    """After the lossy rounds of tests/round_model.py node 0 is declared dead and replica 1 runs for coordinator of
    every group: gpx_election_begin, the PREPAREs at the survivors, gpx_prepare_reply_batch (recorded / elected,
    the carried-over pvalue of the highest ballot per slot, no-ops in the holes, the median), the view change's
    ACCEPTs at the survivors and the new coordinators' rows - with pre-active proposals (some of them duplicates of
    carried requests) and stop requests - against a Python reading of PaxosCoordinator.makeCoordinator /
    handlePrepareReply and PCS:233-263, 271-587 (Candidate in round_model.py)."""
    from tests.round_model import run_rounds
    run_rounds(oracle_lib, G, rounds, seed, p_drop=p_drop, K=K, p_rival=p_rival, p_stop=p_stop, from_disk=seed % 2 == 0,
               failover=True, rounds_after=8)     # ... and eight more rounds under the new coordinators
    assert run_rounds.after > G * 8
    elected, accepts, carried, noops = run_rounds.failover
    assert elected > G // 5 and carried > G // 8 and accepts == (carried + noops) * (K - 1)


@pytest.mark.parametrize("G,rounds,seed,p_drop,K,p_rival,p_stop,failover", [
    (1500, 14, 82, 0.15, 3, 0.03, 0.0, False), (1500, 14, 81, 0.1, 3, 0.0, 0.0, False), (1200, 12, 86, 0.2, 5, 0.0, 0.02, False),
    (1500, 12, 84, 0.1, 4, 0.0, 0.0, True), (1500, 12, 88, 0.25, 3, 0.02, 0.01, True)])
def test_pause_and_hot_restore_between_rounds_against_java_reading(oracle_lib, G, rounds, seed, p_drop, K, p_rival, p_stop, failover):
    """PISM.tryPause of 15 % of the instances of every replica after every round of tests/round_model.py: GPX_S_BUSY
    unless PaxosAcceptor.caughtUp and PaxosCoordinator.caughtUp say so (no pending decision, no accept held in
    memory unless accepts come from disk, no outstanding proposal), else the HotRestoreInfo row - acceptor part and,
    for an active coordinator, ballot / nextProposalSlot / nodeSlotNumbers - and gpx_group_create from that row
    (hotRestore, PISM:677-690), after which the instance plays on like one that never left (odd seeds:
    GET_ACCEPTED_PVALUES_FROM_DISK false, where an acceptor is only caught up once garbage collection has passed).
    After every round also gpx_poke_scan of the coordinators against PISM.pokeLocalCoordinator's reading: the ACCEPT
    for the acceptor's next slot, if the coordinator still waits for it, with the median as it is now and the members
    heard so far."""
    from tests.round_model import run_rounds
    checked, executed = run_rounds(oracle_lib, G, rounds, seed, p_drop=p_drop, K=K, p_rival=p_rival, p_stop=p_stop,
                                   from_disk=seed % 2 == 0, failover=failover, rounds_after=8 if failover else 0, p_pause=0.15, pokes=True,
                                   p_dup_reply=0.3)   # (failover cases: three in ten PREPARE_REPLYs arrive twice)
    assert executed > G * rounds // 5 and run_rounds.busy > G and run_rounds.poked > G
    if seed % 2 == 0:
        assert run_rounds.paused > G and run_rounds.paused_coord > G // 4 and run_rounds.relogged > G // 10
    else:
        assert run_rounds.paused > 0


@pytest.mark.parametrize("K,kw", [(1, dict()), (1, dict(p_stop=0.02, from_disk=False)), (2, dict(p_rival=0.03)),
                                  (2, dict(failover=True, rounds_after=6)), (7, dict(failover=True, rounds_after=4, p_stop=0.01)),
                                  (16, dict(p_rival=0.02))])
def test_whole_round_with_unusual_group_sizes(oracle_lib, K, kw):
    """The round model (pauses and pokes included) for groups of one member (its own vote decides), of two (both
    must vote; after a failure the survivor can never be elected - and is not), of seven, and of sixteen = the
    engine's GPX_KMAX_LIMIT."""
    from tests.round_model import run_rounds
    kw = dict(kw)
    kw.setdefault("from_disk", True)
    checked, executed = run_rounds(oracle_lib, 800, 14, 90 + K, p_drop=0.12, K=K, p_pause=0.1, pokes=True, **kw)
    assert checked > 50_000 and executed > 4000
    if K == 2 and kw.get("failover"):
        assert run_rounds.failover[0] == 0        # nobody elected: one survivor of two is no majority


@pytest.mark.parametrize("base,K,kw", [(2**31 - 6, 3, dict()), (2**31 - 20, 3, dict(p_rival=0.03)), (2**31 - 10, 3, dict(p_stop=0.02, from_disk=False)),
                                       (-2**31 + 1, 4, dict()), (2**31 - 12, 5, dict(p_pause=0.15, pokes=True))])
def test_whole_round_across_the_int_wrap(oracle_lib, base, K, kw):
    """The round model with Java ints (acc_enum_common.I32) for instances restored at slot 1 + base: proposals,
    ACCEPTs, replies with their checkpoint slots, medians, decisions, commits, execution runs, gap scans and PREPAREs
    while the slots cross Integer.MAX_VALUE.  What the Java does there is part of the contract: recordSlotNumber's
    plain < (PCS:809-825) stops recording checkpoint slots once they turn negative, so the medians - and with them
    every acceptedGCSlot - freeze at the wrap; the reading and the oracle agree on that too."""
    from tests.round_model import run_rounds
    kw = dict(kw)
    kw.setdefault("from_disk", True)
    checked, executed = run_rounds(oracle_lib, 1200, 16, 7, p_drop=0.12, K=K, base=base, **kw)
    assert checked > 150_000 and executed > 30_000


@pytest.mark.parametrize("base", [2**31 - 3, 2**31 - 2, 2**31 - 1, -2**31 + 1, 12345])
def test_acceptor_side_at_the_int_wrap_against_java_reading(oracle_lib, base):
    """The acceptor reading with Java ints (tests/acc_enum_common.I32: + and - wrap, comparisons signed; every
    `a - b < 0` of the Java is kept as written) for instances restored at slot 1 + base, so that the slots of the
    ops, the medians, the next slot and the GC slot cross Integer.MAX_VALUE -> MIN_VALUE inside the sequences:
    pairs, and random sequences of 4 and 8 ops, both batch orders."""
    import tests.acc_enum_common as A
    rng = np.random.default_rng(base % 1000)
    n = 0
    for L, count in ((2, None), (4, 8000), (8, 4000)):
        if count is None:
            seqs = [(a, b) for a in A.WIDE[::3] for b in A.WIDE[::5]]
        else:
            seqs = [tuple(A.WIDE[i] for i in row) for row in rng.integers(0, len(A.WIDE), (count, L)).tolist()]
        for order, init in (("interleaved", "create"), ("grouped", "initial")):
            n += A.run_sequences(oracle_lib, seqs, init=init, order=order, base=base)
    assert n > 100_000


def test_acceptor_side_long_random_sequences_against_java_reading(oracle_lib):
    """The same reading over seeded random sequences of 8 and 12 ops per group (the exhaustive plans stop at 4,
    the random ones above at 6)."""
    import tests.acc_enum_common as A
    assert A.run_long_random(oracle_lib, 12_000) > 400_000     # (the GPU test runs 120,000 groups per length)


@pytest.mark.parametrize("K,nprop,init,sample", [(3, 1, [0, 0, 0], None), (3, 2, [1, 0, 2], 60_000), (4, 1, [2, 0, 1, 0], None),
                                                 (5, 1, [0, 2, 1, 0, 3], None), (4, 2, [0, 1, 0, 2], 30_000),
                                                 (3, 3, [2, 1, 0], 30_000)])
def test_pcs_accept_reply_tail_with_checkpoint_slots_enumerated(oracle_lib, K, nprop, init, sample):
    """The enumeration above with recordSlotNumber and a non-trivial median in play: every vote also draws
    its maxCheckpointedSlot from {-1, 0, slot - 1, slot}, nodeSlotNumbers starts non-zero (8^(K * nprop)
    patterns, or a seeded sample of them)."""
    from tests.pcs_enum_common import run_maxcp
    n = run_maxcp(oracle_lib, K, nprop, init, sample=sample, seed=K * 10 + nprop)
    assert n == (sample if sample else 8 ** (K * nprop))
