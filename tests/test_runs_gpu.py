"""Accept replies as a few SORTED RUNS (gigapaxos_amd/csrc/gpx_runs.hip.h): the concatenated replies of the
acceptors, each acceptor's grouped by group, are applied without partitioning the batch - k_runs_check,
k_ar_runs, k_emit_dec_runs, k_merge_runs.  Same answers as the oracle's one-vote-at-a-time replay
(PISM.handleBatchedAcceptReply / handleAcceptReply, PaxosInstanceStateMachine.java:1248-1419 ->
PaxosCoordinatorState.handleAcceptReplyMyBallot / HigherBallot, PCS:597-683), bit for bit: decisions in
order, per-vote status, HotRestoreInfo rows, counters, sampled full state."""
import numpy as np
import pytest

from gigapaxos_amd import (Engine, hri_create, streams, S_OK, S_UNORDERED, D_DECISION, ORDERED_REPLY_RUNS,
                           TRY_REPLY_RUNS)
from tests.parity_common import make_pair, assert_same_state, create_mixed_groups

pytestmark = pytest.mark.gpu


def _same(dh, do, what):
    a, b = dh.as_tuple_array(), do.as_tuple_array()
    assert a.shape == b.shape, f"{what}: {a.shape} vs {b.shape}"
    assert (a == b).all(), f"{what}: first difference at row {int(np.nonzero((a != b).any(1))[0][0])}"
    assert (dh.status == do.status).all(), f"{what}: per-vote status"


@pytest.mark.parametrize("k,mode", [(3, ORDERED_REPLY_RUNS), (3, TRY_REPLY_RUNS), (5, ORDERED_REPLY_RUNS)])
def test_reply_runs_1m_groups_vs_oracle(hip_lib, oracle_lib, k, mode):
    """1 M groups, the votes of a round as K ascending runs: alike (every acceptor answered every group: the
    register fast path), with lost replies (runs differ: searches), with duplicates, stale ballots and a
    sprinkling of higher ballots (preemptions: general replay); under the promise and as a hint."""
    G = 1_000_000
    members = list(range(100, 100 + k))
    eh, eo = make_pair(hip_lib, oracle_lib, 100, G, k, 8, max_batch=G * k + (G * k) // 50 + 4096)
    mem = np.tile(np.array(members, np.int32), (G, 1))
    for e in (eh, eo):
        assert (e.create_groups(np.arange(G), mem, k, hri_create(G, k, 100)) == S_OK).all()
        e.set_ordered_batches(mode)
    g = np.arange(G, dtype=np.int32)
    for r, (mix, drop) in enumerate([(False, 0.0), (False, 0.02), (True, 0.0), (True, 0.01)]):
        for x, y in zip(eh.propose(g), eo.propose(g)):
            assert (x == y).all()
        cols = streams.vote_round_runs(G, members, r, 100, mix=mix, drop=drop)
        dh, do = eh.accept_reply(*cols), eo.accept_reply(*cols)
        _same(dh, do, f"round {r}")
        assert (np.diff(dh.gidx) >= 0).all()
        if not mix and drop == 0.0:
            assert dh.gidx.shape[0] == G and (dh.kind == D_DECISION).all()
    sh, so = eh.snapshot(g)[0], eo.snapshot(g)[0]
    assert sh.tobytes() == so.tobytes()
    assert_same_state(eh, eo, np.random.default_rng(k).integers(0, G, 48))
    assert eh.counters() == eo.counters()
    eh.close()
    eo.close()


def _random_runs(rng, G, nodes, R, slot_lo, slot_hi, my_id, p_absent, hot=None):
    """R ascending runs over random subsets of the groups, random multiplicities (several slots / duplicates
    per group and run), ballots mostly the coordinator's, a few stale and higher ones, acceptors mostly
    members, a few strangers."""
    cols = [[] for _ in range(6)]
    for _ in range(R):
        present = np.nonzero(rng.random(G) >= p_absent)[0].astype(np.int32)
        rep = rng.choice([1, 1, 1, 2, 3], size=present.shape[0])
        gj = np.repeat(present, rep)
        if hot is not None:
            gj = np.sort(np.concatenate([gj, np.full(int(rng.integers(20, 60)), hot, np.int32)]))
        n = gj.shape[0]
        bn = rng.choice([0, 0, 0, 0, 0, 0, 0, 1], size=n).astype(np.int32)
        bc = np.where(rng.random(n) < 0.97, my_id, my_id - 1).astype(np.int32)
        sl = rng.integers(slot_lo, slot_hi + 1, n).astype(np.int32)
        ac = rng.choice(nodes + [nodes[0] - 9], size=n).astype(np.int32)
        cp = (sl - 1 - rng.integers(0, 3, n)).astype(np.int32)
        for c, v in zip(cols, (gj, bn, bc, sl, ac, cp)):
            c.append(v)
    return [np.ascontiguousarray(np.concatenate(c)) for c in cols]


@pytest.mark.parametrize("kmax,G,seed,mode", [(3, 700, 41, ORDERED_REPLY_RUNS), (5, 5000, 42, TRY_REPLY_RUNS),
                                              (8, 300, 43, ORDERED_REPLY_RUNS), (16, 90, 44, TRY_REPLY_RUNS),
                                              (3, 70_000, 45, ORDERED_REPLY_RUNS)])
def test_reply_runs_fuzz(hip_lib, oracle_lib, kmax, G, seed, mode):
    """Runs that are NOT alike: groups absent from the first run (their outputs are parked in later runs: the
    merge), several outstanding slots per group (more outputs than votes in run 0: the merge again), hot
    groups, 1 ... 16 runs, mixed group sizes; then batches that are no few runs at all (17+ runs, an index out
    of range): refused whole under the promise, partitioned under the hint - engine and oracle alike."""
    rng = np.random.default_rng(seed)
    nodes = list(range(100, 100 + max(kmax, 3)))
    eh, eo = make_pair(hip_lib, oracle_lib, 100, G, kmax, 16, max_batch=1 << 21)
    create_mixed_groups(eh, eo, G, kmax, nodes, rng)
    for e in (eh, eo):
        e.set_ordered_batches(mode)
    g = np.arange(G, dtype=np.int32)
    nslots = 0
    for step in range(14):
        newp = int(rng.integers(1, 4))
        for _ in range(newp):                       # one to three more outstanding slots per group
            if nslots < 10:
                for x, y in zip(eh.propose(g), eo.propose(g)):
                    assert (x == y).all()
                nslots += 1
        R = int(rng.choice([1, 2, 3, 3, 5, 9, 16]))
        cols = _random_runs(rng, G, nodes[:kmax], R, max(1, nslots - 3), nslots + 1, 100,
                            p_absent=float(rng.choice([0.0, 0.05, 0.5])), hot=int(rng.integers(0, G)) if step % 3 == 0 else None)
        bad = step in (5, 11)
        if step == 5:                               # too many runs: 20 more descents
            tail = np.tile(np.array([G - 1, 0], np.int32), 20)
            cols = [np.concatenate([cols[0], tail])] + [np.concatenate([c, np.resize(c[:7], 40)]) for c in cols[1:]]
        if step == 11:                              # an index outside the table at the end of the last run
            cols[0] = cols[0].copy()
            cols[0][-1] = G + 3
        dh, do = eh.accept_reply(*cols), eo.accept_reply(*cols)
        _same(dh, do, f"step {step} R={R}")
        if bad and mode == ORDERED_REPLY_RUNS:
            assert (dh.status == S_UNORDERED).all() and dh.gidx.shape[0] == 0
        else:
            assert (np.diff(dh.gidx) >= 0).all()
    assert_same_state(eh, eo, range(min(G, 400)))
    assert eh.snapshot(g)[0].tobytes() == eo.snapshot(g)[0].tobytes()
    assert eh.counters() == eo.counters()
    eh.close()
    eo.close()


def test_reply_runs_identical_to_partition_path(hip_lib):
    """The same batches through the runs path and through the partition pipeline on two HIP engines: every
    output and the whole state identical (what gpx.h promises: results do not depend on the path)."""
    rng = np.random.default_rng(9)
    G, k = 20_000, 3
    members = [100, 101, 102]
    ea = Engine(hip_lib, 100, G, kmax=k, window=8, max_batch=1 << 20)
    eb = Engine(hip_lib, 100, G, kmax=k, window=8, max_batch=1 << 20)
    mem = np.tile(np.array(members, np.int32), (G, 1))
    for e in (ea, eb):
        assert (e.create_groups(np.arange(G), mem, k, hri_create(G, k, 100)) == S_OK).all()
    ea.set_ordered_batches(TRY_REPLY_RUNS)
    g = np.arange(G, dtype=np.int32)
    for r in range(6):
        for x, y in zip(ea.propose(g), eb.propose(g)):
            assert (x == y).all()
        cols = streams.vote_round_runs(G, members, r, 100, mix=r % 2 == 1, drop=0.03 * (r % 3))
        if r == 4:                                  # a shuffled batch: the hint's fallback
            p = rng.permutation(cols[0].shape[0])
            cols = [np.ascontiguousarray(c[p]) for c in cols]
        da, db = ea.accept_reply(*cols), eb.accept_reply(*cols)
        _same(da, db, f"round {r}")
    assert ea.snapshot(g)[0].tobytes() == eb.snapshot(g)[0].tobytes()
    assert ea.counters() == eb.counters()
    ea.close()
    eb.close()
