"""Parity tests proper: the hand-written HIP engine (libgpx_hip.so, through the C-ABI) against the
CPU oracle on identical seeded inputs.  Bit-exact: every output column, the decided
(group, slot, ballot, medianCheckpointedSlot) stream in order, and the full per-group state."""
import numpy as np
import pytest

from gigapaxos_amd import (Engine, hri_create, hri_initial, make_hri, streams, S_OK, S_WINDOW,
                           S_NOGROUP, D_DECISION, D_PREEMPTED)
from gigapaxos_amd.loopback import LoopbackCluster
from tests.parity_common import (make_pair, create_mixed_groups, fuzz, assert_same_state, wrap32,
                                 churn_run)

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["partition path", "sorted-runs hint"])
def _accept_reply_path(request, monkeypatch):
    """Accept-reply batches go through the partition pipeline or, with the GPX_TRY_REPLY_RUNS hint (here set
    for every engine through the test switch GPX_TRY_RUNS=1, read at engine creation), first through the
    sorted-runs check (gpx_runs.hip.h): a batch that happens to be a few ascending runs is applied without
    partition, any other falls back behind it - every case of this file runs both ways."""
    monkeypatch.setenv("GPX_TRY_RUNS", "1" if request.param.startswith("sorted") else "0")


NODES = [100, 101, 102, 103, 104, 105, 106, 107]


@pytest.mark.parametrize("kmax,seed", [(3, 1), (5, 2), (8, 3), (16, 4)])
def test_fuzz_mixed_ops(hip_lib, oracle_lib, kmax, seed):
    rng = np.random.default_rng(seed)
    nodes = NODES if kmax <= 8 else list(range(100, 120))
    G = 64
    eh, eo = make_pair(hip_lib, oracle_lib, 100, G, kmax, 64)
    create_mixed_groups(eh, eo, G, kmax, nodes, rng)
    fuzz(eh, eo, G, nodes, rng, steps=250, batch=300)


@pytest.mark.parametrize("kmax,G,seed", [(3, 48, 21), (5, 700, 22), (3, 3000, 23)])
def test_fuzz_ordered_batches(hip_lib, oracle_lib, kmax, G, seed):
    """Batches grouped by group (gidx non-decreasing): ACCEPT and COMMIT batches take the direct path
    (gpx_direct.hip.h, no partition), runs of several records per group replayed in array order."""
    import os
    if G == 3000 and os.environ.get("GPX_TRY_RUNS") == "1":
        pytest.skip("the largest case runs once: ten seconds of the GPU suite, and its vote batches are shuffled - the hint changes nothing")
    rng = np.random.default_rng(seed)
    eh, eo = make_pair(hip_lib, oracle_lib, 100, G, kmax, 64)
    create_mixed_groups(eh, eo, G, kmax, NODES, rng)
    fuzz(eh, eo, G, NODES, rng, steps=160, batch=max(300, G), ordered=True)


def test_ordered_batches_promise(hip_lib, oracle_lib):
    """gpx_engine_set_ordered_batches: with the promise only the direct path is launched; a batch
    that keeps it gives the usual answers, one that breaks it is applied up to its first violation and refused from
    there on (GPX_S_UNORDERED, no state change) - engine and oracle alike."""
    from gigapaxos_amd import ORDERED_PROPOSE, ORDERED_ACCEPT, ORDERED_COMMIT, S_UNORDERED
    rng = np.random.default_rng(31)
    G, kmax = 900, 3
    eh, eo = make_pair(hip_lib, oracle_lib, 100, G, kmax, 64)
    create_mixed_groups(eh, eo, G, kmax, NODES, rng)
    for e in (eh, eo):
        e.set_ordered_batches(ORDERED_ACCEPT | ORDERED_COMMIT)
    fuzz(eh, eo, G, NODES, rng, steps=120, batch=1200, ordered=True)   # 1 batch in 8 carries a bad index
    for e in (eh, eo):
        e.set_ordered_batches(ORDERED_PROPOSE | ORDERED_ACCEPT | ORDERED_COMMIT)
    for g in (np.arange(G, dtype=np.int32), np.arange(0, G, 7, dtype=np.int32),
              np.array([5, 5, 9], np.int32), np.arange(G, dtype=np.int32)[::-1].copy(),
              np.array([3, 900, 901], np.int32), np.array([7], np.int32)):
        ra, rb = eh.propose(g), eo.propose(g)
        for x, y, nm in zip(ra, rb, ("slot", "bnum", "bcoord", "median", "status")):
            assert x.tolist() == y.tolist(), nm
        bad = np.nonzero((g < 0) | (g >= G) | np.concatenate([[False], np.diff(g) <= 0]))[0]
        v = int(bad[0]) if bad.shape[0] else g.shape[0]    # the first violation: refused from there on, applied before
        assert (ra[4][v:] == S_UNORDERED).all() and not (ra[4][:v] == S_UNORDERED).any()
    z = np.zeros(4, np.int32)
    res = []
    for e in (eh, eo):   # broken promise on the acceptor side: the records before the first violation are applied
        (rb_, rc_, rm_, rf_, st), runs = e.accept(np.array([4, 2, 2, 9], np.int32), z, np.full(4, 100, np.int32), z + 1, z)
        assert (st[1:] == S_UNORDERED).all() and st[0] != S_UNORDERED and not rb_[1:].any() and not rf_[1:].any()
        st2, runs2 = e.commit(np.array([4, 4, 2, 9], np.int32), z, np.full(4, 100, np.int32), z + 1, z)
        assert (st2[2:] == S_UNORDERED).all() and not (st2[:2] == S_UNORDERED).any()
        res.append([x.tolist() for x in (rb_, rc_, rm_, rf_, st, runs.as_tuple_array(), st2, runs2.as_tuple_array())])
    assert res[0] == res[1]
    assert_same_state(eh, eo, range(G))
    assert eh.counters() == eo.counters()


def test_fuzz_wraparound(hip_lib, oracle_lib):
    """slots straddle Integer.MAX_VALUE -> MIN_VALUE."""
    rng = np.random.default_rng(5)
    base = (1 << 31) - 15
    eh, eo = make_pair(hip_lib, oracle_lib, 100, 32, 3, 64)
    create_mixed_groups(eh, eo, 32, 3, NODES[:4], rng, slot_base=base)
    fuzz(eh, eo, 32, NODES[:4], rng, steps=200, batch=150, slot_base=base)


def test_fuzz_accepts_in_memory(hip_lib, oracle_lib):
    """GET_ACCEPTED_PVALUES_FROM_DISK = false (logging disabled): accepts stay after execution."""
    rng = np.random.default_rng(6)
    eh, eo = make_pair(hip_lib, oracle_lib, 100, 32, 3, 64, flags=0)
    create_mixed_groups(eh, eo, 32, 3, NODES[:4], rng)
    fuzz(eh, eo, 32, NODES[:4], rng, steps=200, batch=150)


def _same_decisions(da, db):
    assert da.as_tuple_array().tolist() == db.as_tuple_array().tolist()
    assert da.status.tolist() == db.status.tolist()


@pytest.mark.parametrize("kmax,seed", [(3, 51), (4, 52), (5, 53), (8, 54)])
def test_steady_state_votes_fast_path(hip_lib, oracle_lib, kmax, seed):
    """Batches in which every group's votes answer ONE outstanding slot at the current ballot take a
    straight-line replay in k_bucket_ar16 (a whole wave at a time; anything else sends the wave down
    apply_ar_group).  One to ten votes per group - members in any order, the same member twice, a
    node that is no member, varying maxCheckpointedSlot - for the newest or an older outstanding slot,
    groups of different sizes, rounds with a few votes of another ballot mixed in (their waves take
    the general path): decisions in order, statuses, HotRestoreInfo rows and counters as the oracle
    (tests/parity_common.steady_state_run; scripts/soak.py runs it with fresh seeds)."""
    from tests.parity_common import steady_state_run
    steady_state_run(hip_lib, oracle_lib, kmax, seed, NODES)


@pytest.mark.parametrize("nvotes", [17, 600, 4096, 20000])
def test_hot_group_long_segments(hip_lib, oracle_lib, nvotes):
    """All votes of a batch hit ONE group (a BATCHED_ACCEPT_REPLY with many slots):
    exercises the long-segment arrival-order sort (LDS and global-memory variants)."""
    rng = np.random.default_rng(nvotes)
    W = 64
    eh, eo = make_pair(hip_lib, oracle_lib, 100, 4, 3, W, max_batch=1 << 15)
    mem = np.tile(np.array([100, 101, 102], np.int32), (4, 1))
    for e in (eh, eo):
        assert (e.create_groups(np.arange(4), mem, 3, hri_create(4, 3, 100)) == S_OK).all()
    nprop = 40
    g = np.full(nprop, 2, np.int32)
    for x, y in zip(eh.propose(g), eo.propose(g)):
        assert x.tolist() == y.tolist()
    slot = rng.integers(1, nprop + 1, nvotes).astype(np.int32)
    acc = rng.choice([100, 101, 102, 55], size=nvotes).astype(np.int32)
    mcp = rng.integers(-1, 30, nvotes).astype(np.int32)
    bnum = (rng.random(nvotes) < 0.002).astype(np.int32)
    gv = np.full(nvotes, 2, np.int32)
    gv[rng.random(nvotes) < 0.05] = 1
    args = (gv, bnum, np.full(nvotes, 100, np.int32), slot, acc, mcp)
    _same_decisions(eh.accept_reply(*args), eo.accept_reply(*args))
    assert_same_state(eh, eo, range(4))


def test_window_overflow_statuses(hip_lib, oracle_lib):
    """The engine tracks W slots per group per map; beyond that a record is DROPPED with
    GPX_S_WINDOW and no state changes.  Feeding the oracle only the accepted records must give
    the same state."""
    W = 4
    eh, eo = make_pair(hip_lib, oracle_lib, 100, 2, 3, W)
    mem = np.tile(np.array([100, 101, 102], np.int32), (2, 1))
    for e in (eh, eo):
        e.create_groups(np.arange(2), mem, 3, hri_create(2, 3, 100))
    g = np.zeros(7, np.int32)
    sl, bn, bc, md, st = eh.propose(g)
    assert st.tolist() == [S_OK] * 4 + [S_WINDOW] * 3 and sl[:4].tolist() == [1, 2, 3, 4]
    eo.propose(g[:4])
    assert_same_state(eh, eo, range(2))
    # decide slot 1 -> one more proposal fits
    v = (np.zeros(2, np.int32), np.zeros(2, np.int32), np.full(2, 100, np.int32),
         np.ones(2, np.int32), np.array([100, 101], np.int32), np.zeros(2, np.int32))
    _same_decisions(eh.accept_reply(*v), eo.accept_reply(*v))
    sl, _, _, _, st = eh.propose(g[:2])
    assert st.tolist() == [S_OK, S_WINDOW] and sl[0] == 5
    eo.propose(g[:1])
    assert_same_state(eh, eo, range(2))
    # acceptor side: commits further ahead than W, accepts colliding in the ring
    slots = np.array([1, 2, 7, 5, 9], np.int32)  # 5 and 9 collide with live slot 1 in a W=4 ring
    z = np.zeros(5, np.int32)
    (rb, rc, rm, rf, st), _ = eh.accept(np.ones(5, np.int32), z, np.full(5, 100, np.int32), slots, z)
    assert st.tolist() == [S_OK, S_OK, S_OK, S_WINDOW, S_WINDOW]
    (_, _, _, _, sto), _ = eo.accept(np.ones(5, np.int32), z, np.full(5, 100, np.int32), slots, z)
    assert sto.tolist() == st.tolist()  # the oracle keeps the same ring rule (round 3)
    st, _ = eh.commit(np.ones(3, np.int32), z[:3], np.full(3, 100, np.int32),
                      np.array([5, 4, 2], np.int32), z[:3])
    assert st.tolist() == [S_WINDOW, S_OK, S_OK]
    sto, _ = eo.commit(np.ones(3, np.int32), z[:3], np.full(3, 100, np.int32), np.array([5, 4, 2], np.int32), z[:3])
    assert sto.tolist() == st.tolist()
    assert_same_state(eh, eo, range(2))


def test_config1_loopback_one_group(hip_lib):
    """BASELINE config #1 on the engine: 3 replicas, 1 group, in-order execution everywhere."""
    c = LoopbackCluster(hip_lib, [100, 101, 102], 1, window=8, max_batch=1024)
    n = 300
    for r in range(n):
        dec = c.round([0])
        assert dec[0, :4].tolist() == [0, r + 1, 0, 100] and dec[0, 5] == D_DECISION
    for nid in (100, 101, 102):
        ex = c.executed(nid)
        slots = np.concatenate([np.arange(f, f + cnt) for _, f, cnt in ex])
        assert slots.tolist() == list(range(1, n + 1))


def test_config2_10k_groups_full_pipeline(hip_lib, oracle_lib):
    """BASELINE config #2: 10k groups, 3 replicas, propose -> accept x3 -> reply x3 -> decide ->
    commit x3 -> exec, one batch per round; coordinators spread over the replicas as
    roundRobinCoordinator(name_g) would.  Engine cluster vs oracle cluster, identical streams."""
    G, R = 10000, 12
    rng = np.random.default_rng(2)
    coord = rng.choice([100, 101, 102], size=G).astype(np.int32)
    ch = LoopbackCluster(hip_lib, [100, 101, 102], G, window=8, max_batch=1 << 16, coordinator=coord)
    co = LoopbackCluster(oracle_lib, [100, 101, 102], G, window=8, coordinator=coord)
    for r in range(R):
        groups = rng.permutation(G).astype(np.int32)
        dh, do = ch.round(groups), co.round(groups)
        assert dh.tolist() == do.tolist()
        assert dh.shape[0] == G
    for nid in (100, 101, 102):
        assert ch.executed(nid).tolist() == co.executed(nid).tolist()
        eh, eo = ch.engines[nid], co.engines[nid]
        sh, so = eh.snapshot(np.arange(G))[0], eo.snapshot(np.arange(G))[0]
        assert sh.tobytes() == so.tobytes()
        assert (sh["acc_slot"] == R + 1).all()
        assert_same_state(eh, eo, rng.integers(0, G, 40))


@pytest.mark.parametrize("k,mix,shuffled", [(3, False, True), (3, True, True), (3, True, False), (5, True, True)])
def test_config3_vote_stream(hip_lib, oracle_lib, k, mix, shuffled):
    """BASELINE config #3/#4 stream (synthetic accept-reply stream, engine = coordinator of every
    group) at a size the oracle finishes in seconds: decided stream identical, in order."""
    G, R = 1 << 16, 6
    members = list(range(100, 100 + k))
    eh, eo = make_pair(hip_lib, oracle_lib, 100, G, k, 8, max_batch=1 << 19)
    mem = np.tile(np.array(members, np.int32), (G, 1))
    for e in (eh, eo):
        assert (e.create_groups(np.arange(G), mem, k, hri_create(G, k, 100)) == S_OK).all()
    for r in range(R):
        g = np.arange(G, dtype=np.int32)
        for x, y in zip(eh.propose(g), eo.propose(g)):
            assert (x == y).all()
        cols = streams.vote_round(G, members, r, 100, config_id=3 if k == 3 else 4, shuffled=shuffled, mix=mix)
        dh, do = eh.accept_reply(*cols), eo.accept_reply(*cols)
        assert (dh.as_tuple_array() == do.as_tuple_array()).all()
        assert dh.as_tuple_array().shape == do.as_tuple_array().shape
        assert (dh.status == do.status).all()
        if not mix:
            assert dh.gidx.shape[0] == G and (dh.kind == D_DECISION).all()
    sh, so = eh.snapshot(np.arange(G))[0], eo.snapshot(np.arange(G))[0]
    assert sh.tobytes() == so.tobytes()
    assert_same_state(eh, eo, np.random.default_rng(0).integers(0, G, 64))
    assert eh.counters() == eo.counters()


def test_full_size_properties_1m_groups(hip_lib):
    """BASELINE config #3 at FULL size (1 M groups, 3 M votes per round): size-independent
    properties instead of the oracle — exactly one DECISION per group per clean round, each at the
    round's slot with median = round (createHRI rows), output grouped by gidx ascending, and the
    state rows advance uniformly."""
    G, R = 1_000_000, 3
    e = Engine(hip_lib, 100, G, kmax=3, window=8, max_batch=3 * G + 65536)
    mem = np.tile(np.array([100, 101, 102], np.int32), (G, 1))
    assert (e.create_groups(np.arange(G), mem, 3, hri_create(G, 3, 100)) == S_OK).all()
    for r in range(R):
        sl, bn, bc, md, st = e.propose(np.arange(G, dtype=np.int32))
        assert (st == S_OK).all() and (sl == r + 1).all() and (md == max(r - 1, 0)).all()
        cols = streams.vote_round(G, [100, 101, 102], r, 100)
        d = e.accept_reply(*cols)
        assert d.gidx.shape[0] == G and (d.kind == D_DECISION).all()
        assert (d.slot == r + 1).all() and (d.median_cp == r).all()
        assert (np.bincount(d.gidx, minlength=G) == 1).all()
        # output order contract: grouped by gidx ascending (one decision per group here)
        assert (d.gidx == np.arange(G, dtype=np.int32)).all()
    rows, st = e.snapshot(np.arange(G))
    assert (rows["next_proposal_slot"] == R + 1).all()
    assert (rows["node_slots"][:, :3] == R - 1).all()
    assert e.counters()[:2] == (3 * G * R, G * R)


def test_config5_churn_create_retire_mid_run(hip_lib, oracle_lib):
    """BASELINE config #5 at a size the oracle finishes in seconds: group create / delete mid-run,
    gidx rows reused, late votes for retired groups dropped — decided stream, per-record statuses,
    HotRestoreInfo rows of the retired groups and the final state identical to the oracle."""
    G_live, cap, R, k = 20000, 20000 + 3 * 200, 8, 5
    eh, eo = make_pair(hip_lib, oracle_lib, 100, cap, k, 8, max_batch=1 << 18)
    (oh, oo), live = churn_run([eh, eo], G_live, cap, R, k, seed=5)
    for r, (a, b) in enumerate(zip(oh, oo)):
        for x, y, nm in zip(a, b, ("decisions", "vote status", "propose out", "propose status",
                                    "retired rows", "retire status", "create status")):
            if isinstance(x, bytes):
                assert x == y, f"round {r} {nm}"
            else:
                assert x.shape == y.shape and (x == y).all(), f"round {r} {nm}"
        dec, vst = a[0], a[1]
        assert (a[5] == S_OK).all() and (a[6] == S_OK).all()
        assert dec.shape[0] == G_live          # every live group decides every round
        if r > 0:
            assert (vst == S_NOGROUP).sum() == k * max(1, int(G_live * 0.01))
    sh, so = eh.snapshot(np.arange(cap))[0], eo.snapshot(np.arange(cap))[0]
    assert sh.tobytes() == so.tobytes()
    assert_same_state(eh, eo, np.random.default_rng(1).integers(0, cap, 64))
    assert eh.counters() == eo.counters()


@pytest.mark.parametrize("order", ["ascending", "ascending_sparse", "duplicates", "descending", "out_of_range"])
def test_propose_batch_orders(hip_lib, oracle_lib, order):
    """The proposal path picks its back end on the device: a strictly ascending in-range gidx column
    (one record per group) is applied directly, anything else is regrouped first.  Both must give
    the oracle's answer, including stop requests, non-existent / stopped groups and forwards."""
    G, k = 5000, 3
    rng = np.random.default_rng(11)
    eh, eo = make_pair(hip_lib, oracle_lib, 100, G, k, 32, max_batch=1 << 15)  # window 32: never fills here
    create_mixed_groups(eh, eo, G - 500, k, NODES[:5], rng)  # rows G-500.. stay non-existent
    for rnd in range(4):
        if order == "ascending":
            g = np.arange(G, dtype=np.int32)
        elif order == "ascending_sparse":
            g = np.sort(rng.choice(G, size=G // 7, replace=False)).astype(np.int32)
        elif order == "duplicates":
            g = np.sort(rng.integers(0, G, G // 2)).astype(np.int32)
        elif order == "descending":
            g = np.arange(G, dtype=np.int32)[::-1].copy()
        else:
            g = np.arange(-1, G + 1, dtype=np.int32)
        stop = (rng.random(g.shape[0]) < (0.02 if rnd == 2 else 0.0)).astype(np.uint8)
        ra, rb = eh.propose(g, stop), eo.propose(g, stop)
        for x, y, nm in zip(ra, rb, ("slot", "bnum", "bcoord", "median", "status")):
            assert x.tolist() == y.tolist(), f"round {rnd} {nm}"
    sh, so = eh.snapshot(np.arange(G))[0], eo.snapshot(np.arange(G))[0]
    assert sh.tobytes() == so.tobytes()
    assert_same_state(eh, eo, rng.integers(0, G, 48))
    assert eh.counters() == eo.counters()


@pytest.mark.parametrize("K,nprop", [(3, 2), (5, 2)])
def test_pcs_main_accept_reply_tail_every_coin_on_engine(hip_lib, K, nprop):
    """The same enumeration (PaxosCoordinatorState.java:1173-1213, every coin) on the HIP engine."""
    from tests.pcs_enum_common import run_all
    assert run_all(hip_lib, K, nprop) == 1 << (K * nprop)
