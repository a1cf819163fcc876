"""Ordered batches of more than 65,536 records (gpx_one.hip.h, round 4): a check kernel leaves the batch's first
violation, ONE work kernel applies everything before it and refuses the rest, and the compaction pass of an unusual
batch runs only when it is needed (GPX_LAZY_OUTPUTS / gpx_compact_last_dev).  Engine against the oracle."""
import numpy as np
import pytest

from gigapaxos_amd import (Engine, hri_create, S_OK, S_UNORDERED, ORDERED_PROPOSE, ORDERED_ACCEPT, ORDERED_COMMIT,
                           ORDERED_REPLY_RUNS, LAZY_OUTPUTS, C_HASVALUE, streams)
from tests.parity_common import make_pair, assert_same_state, create_mixed_groups, fuzz

pytestmark = pytest.mark.gpu
NODES = [100, 101, 102]


@pytest.mark.parametrize("seed,G,batch,steps", [(41, 6_000, 100_000, 3), (42, 8_000, 150_000, 2)])
def test_large_ordered_batches_under_the_promise(hip_lib, oracle_lib, seed, G, batch, steps):
    """parity_common.fuzz with grouped batches of up to `batch` records under PROPOSE | ACCEPT | COMMIT: k_propose_one /
    k_ac_one for the batches above 65,536 records, k_*_small below; one batch in eight carries an index out of range
    (refused from there on), runs of several records per group, commits that execute nothing or several slots."""
    rng = np.random.default_rng(seed)
    eh, eo = make_pair(hip_lib, oracle_lib, 100, G, 3, 64, max_batch=batch + 64)
    create_mixed_groups(eh, eo, G, 3, NODES, rng)
    for e in (eh, eo):
        e.set_ordered_batches(ORDERED_PROPOSE | ORDERED_ACCEPT | ORDERED_COMMIT)
    fuzz(eh, eo, G, NODES, rng, steps=steps, batch=batch, ordered=True)
    assert_same_state(eh, eo, rng.integers(0, G, 400))
    assert eh.counters() == eo.counters()
    eh.close(), eo.close()


@pytest.mark.parametrize("where", ["descent", "out of range", "repeated group", "first record", "last record"])
def test_broken_promise_refuses_from_the_first_violation(hip_lib, oracle_lib, where):
    """One violation planted in a batch of 300,000 grouped records: everything before it is applied, everything from it
    on carries GPX_S_UNORDERED and zero outputs - ACCEPT, COMMIT and PROPOSE, engine and oracle alike."""
    G, n = 200_000, 300_000
    rng = np.random.default_rng(len(where))
    eh, eo = make_pair(hip_lib, oracle_lib, 100, G, 3, 8, max_batch=n + 64)
    mem = np.tile(np.array(NODES, np.int32), (G, 1))
    for e in (eh, eo):
        assert (e.create_groups(np.arange(G), mem, 3, hri_create(G, 3, 100)) == S_OK).all()
        e.set_ordered_batches(ORDERED_PROPOSE | ORDERED_ACCEPT | ORDERED_COMMIT)
    g = np.sort(rng.integers(0, G, n)).astype(np.int32)
    v = {"first record": 0, "last record": n - 1}.get(where, int(rng.integers(70_000, n - 70_000)))
    if where == "descent":
        g[v] = g[v - 1] - 1
    elif where == "repeated group":      # the group of record v - 3000 comes again after other groups
        g[v] = g[v - 3000]
        assert g[v] < g[v - 1]
    else:
        g[v] = G + 3 if where != "first record" else -1
    first = int(np.nonzero((g < 0) | (g >= G) | np.concatenate([[False], np.diff(g) < 0]))[0][0])
    # ACCEPT: slots 1, 2, .. per group in array order
    slot = np.ones(n, np.int32)
    same = np.concatenate([[False], g[1:] == g[:-1]])
    run = np.zeros(n, np.int64)
    for i in np.nonzero(same)[0]:
        run[i] = run[i - 1] + 1
    slot += run.astype(np.int32)
    z = np.zeros(n, np.int32)
    bc = np.full(n, 100, np.int32)
    out = []
    for e in (eh, eo):
        (rb, rc, rm, rf, st), runs = e.accept(g, z, bc, slot, z)
        assert (st[first:] == S_UNORDERED).all() and not (st[:first] == S_UNORDERED).any()
        assert not rb[first:].any() and not rc[first:].any() and not rf[first:].any()
        st2, runs2 = e.commit(g, z, bc, slot, z)
        assert (st2[first:] == S_UNORDERED).all() and not (st2[:first] == S_UNORDERED).any()
        gp = np.unique(g[(g >= 0) & (g < G)])
        if where == "descent":
            gp = np.concatenate([gp[:1000], gp[999:2000]])       # a non-ascent for the strict promise
        pr = e.propose(gp.astype(np.int32))
        out.append([x.tolist() for x in (rb, rc, rm, rf, st, runs.as_tuple_array(), st2, runs2.as_tuple_array()) + tuple(pr)])
    assert out[0] == out[1]
    assert_same_state(eh, eo, np.concatenate([rng.integers(0, G, 300), g[max(0, v - 3):v + 3].clip(0, G - 1)]))
    assert eh.counters() == eo.counters()
    eh.close(), eo.close()


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("G,exchange", [(200_000, True), (20_000, True), (20_000, False)])
def test_lazy_outputs_on_the_device_path(hip_lib, oracle_lib, monkeypatch, G, exchange):
    """GPX_LAZY_OUTPUTS through the *_dev calls: a usual batch comes back dense with its count and no compaction kernel
    runs (the engine's launch profile says so); an unusual one comes back with a negative count and
    gpx_compact_last_dev makes it the oracle's.  20,000 records: one launch, k_ac_pers (a grid that is resident for sure,
    its workgroups exchange the verdict among themselves); 200,000 records, or an engine that may not count on its grid
    being resident at once (here: GPX_XCHG_SLOTS=0; in production: many streams on the device): k_one_check + k_ac_one."""
    import torch
    if not exchange:
        monkeypatch.setenv("GPX_XCHG_SLOTS", "0")
    eh, eo = make_pair(hip_lib, oracle_lib, 100, G, 3, 8, max_batch=3 * G + 64)
    mem = np.tile(np.array(NODES, np.int32), (G, 1))
    for e in (eh, eo):
        assert (e.create_groups(np.arange(G), mem, 3, hri_create(G, 3, 100)) == S_OK).all()
    eh.set_ordered_batches(ORDERED_PROPOSE | ORDERED_ACCEPT | ORDERED_COMMIT | ORDERED_REPLY_RUNS | LAZY_OUTPUTS)
    eo.set_ordered_batches(ORDERED_PROPOSE | ORDERED_ACCEPT | ORDERED_COMMIT | ORDERED_REPLY_RUNS)
    g = np.arange(G, dtype=np.int32)
    z, bc = np.zeros(G, np.int32), np.full(G, 100, np.int32)
    i32 = lambda n: torch.zeros(n, dtype=torch.int32, device="cuda")  # noqa: E731
    u8 = lambda n: torch.zeros(n, dtype=torch.uint8, device="cuda")  # noqa: E731
    P = lambda t: t.data_ptr()  # noqa: E731

    def accept_dev(gg, slot, median):
        n = gg.shape[0]
        cols = [_dev(torch, c) for c in (gg, np.zeros(n, np.int32), np.full(n, 100, np.int32), slot, median)]
        fl = u8(n)
        o = [i32(n) for _ in range(3)] + [u8(n), u8(n)] + [i32(n) for _ in range(3)] + [i32(1)]
        eh.call_dev("accept_batch", n, *[P(c) for c in cols], P(fl), *[P(t) for t in o])
        torch.cuda.synchronize()
        return o

    def commit_dev(gg, slot, median, kind):
        n = gg.shape[0]
        cols = [_dev(torch, c) for c in (gg, np.zeros(n, np.int32), np.full(n, 100, np.int32), slot, median)]
        o = [u8(n)] + [i32(n) for _ in range(3)] + [i32(1)]
        eh.call_dev("commit_batch", n, *[P(c) for c in cols], P(_dev(torch, kind)), *[P(t) for t in o])
        torch.cuda.synchronize()
        return o

    def runs_of(o, k):
        m = int(o[-1].item())
        return np.stack([o[k].cpu().numpy()[:m], o[k + 1].cpu().numpy()[:m], o[k + 2].cpu().numpy()[:m]], axis=1)

    eh.profile(2)
    # 1) usual ACCEPT batch: nothing released, count 0, no compaction launched
    o = accept_dev(g, np.ones(G, np.int32), z)
    (rb, rc, rm, rf, st), runs = eo.accept(g, z, bc, np.ones(G, np.int32), z)
    assert int(o[-1].item()) == 0 and runs.gidx.shape[0] == 0
    assert (o[0].cpu().numpy() == rb).all() and (o[2].cpu().numpy() == rm).all() and (o[4].cpu().numpy() == st).all()
    # 2) usual COMMIT batch: one run per record, dense as parked
    kind = np.full(G, C_HASVALUE, np.uint8)
    o = commit_dev(g, np.ones(G, np.int32), z, kind)
    st2, runs2 = eo.commit(g, z, bc, np.ones(G, np.int32), z, kind)
    assert int(o[-1].item()) == G and (runs_of(o, 1) == runs2.as_tuple_array()).all() and (o[0].cpu().numpy() == st2).all()
    prof = eh.profile_read()
    assert "k_emit_runs_direct" not in prof and "k_copy_runs" not in prof and "k_order_check" not in prof, prof
    # one launch while the grid is resident for sure (2 workgroups per CU: 131,072 records on an MI355X), else check + work
    assert (sorted(prof) == ["k_ac_pers"]) if (exchange and G <= 192 * 256) else ("k_ac_one" in prof and "k_one_check" in prof), prof
    # 3) unusual COMMIT batch: slot 3 before slot 2 for a third of the groups (executes nothing), slot 2 for the rest
    sl = np.where(g % 3 == 0, 3, 2).astype(np.int32)
    o = commit_dev(g, sl, z, kind)
    st3, runs3 = eo.commit(g, z, bc, sl, z, kind)
    assert int(o[-1].item()) < 0
    eh.compact_last_dev()
    torch.cuda.synchronize()
    assert int(o[-1].item()) == runs3.gidx.shape[0] and (runs_of(o, 1) == runs3.as_tuple_array()).all()
    assert (o[0].cpu().numpy() == st3).all()
    # 4) ... and the missing slot 2 arrives for those groups: two slots execute per commit
    gg = g[g % 3 == 0]
    o = commit_dev(gg, np.full(gg.shape[0], 2, np.int32), np.zeros(gg.shape[0], np.int32), kind[:gg.shape[0]])
    st4, runs4 = eo.commit(gg, np.zeros(gg.shape[0], np.int32), np.full(gg.shape[0], 100, np.int32),
                           np.full(gg.shape[0], 2, np.int32), np.zeros(gg.shape[0], np.int32), kind[:gg.shape[0]])
    assert int(o[-1].item()) == gg.shape[0]   # one run per record: REGULAR whatever the run's length
    assert (runs_of(o, 1) == runs4.as_tuple_array()).all() and (runs4.count == 2).all()
    # 5) unusual ACCEPT batch: placeholders (commits without value) wait for their ACCEPTs, which then release them
    ph = np.zeros(G, np.uint8)
    o = commit_dev(g, np.full(G, 4, np.int32), z, ph)
    st5, runs5 = eo.commit(g, z, bc, np.full(G, 4, np.int32), z, ph)
    assert int(o[-1].item()) < 0
    eh.compact_last_dev()
    torch.cuda.synchronize()
    assert int(o[-1].item()) == 0 and runs5.gidx.shape[0] == 0
    o = accept_dev(g, np.full(G, 4, np.int32), z)
    (rb, rc, rm, rf, st6), runs6 = eo.accept(g, z, bc, np.full(G, 4, np.int32), z)
    assert int(o[-1].item()) < 0
    eh.compact_last_dev()
    torch.cuda.synchronize()
    assert (runs_of(o, 5) == runs6.as_tuple_array()).all() and runs6.gidx.shape[0] > G // 4
    assert (o[4].cpu().numpy() == st6).all() and (o[3].cpu().numpy() == rf).all()
    assert_same_state(eh, eo, np.random.default_rng(5).integers(0, G, 300))
    assert eh.counters() == eo.counters()
    eh.close(), eo.close()


@pytest.mark.parametrize("K,G,exchange", [(3, 150_000, True), (5, 150_000, True), (3, 20_000, True), (5, 9_000, True),
                                          (3, 20_000, False)])
def test_lazy_reply_runs(hip_lib, oracle_lib, monkeypatch, K, G, exchange):
    """Accept replies as K ascending runs under ORDERED_REPLY_RUNS | LAZY_OUTPUTS: the regular round's count is
    published without a compaction launch (by k_runs_check; calls of at most 65,536 votes: by the ONE kernel they
    take, k_ar_runs<.., SMALL>); a round with lost votes comes back negative and is compacted on demand."""
    import torch
    if not exchange:   # the grid may not be resident at once: k_runs_check + k_ar_runs, as for larger calls
        monkeypatch.setenv("GPX_XCHG_SLOTS", "0")
    members = list(range(100, 100 + K))
    eh, eo = make_pair(hip_lib, oracle_lib, 100, G, K, 8, max_batch=K * G + 64)
    mem = np.tile(np.array(members, np.int32), (G, 1))
    for e in (eh, eo):
        assert (e.create_groups(np.arange(G), mem, K, hri_create(G, K, 100)) == S_OK).all()
    eh.set_ordered_batches(ORDERED_PROPOSE | ORDERED_REPLY_RUNS | LAZY_OUTPUTS)
    eo.set_ordered_batches(ORDERED_PROPOSE | ORDERED_REPLY_RUNS)
    g = np.arange(G, dtype=np.int32)
    P = lambda t: t.data_ptr()  # noqa: E731
    rng = np.random.default_rng(K)
    for r in range(4):
        for x, y in zip(eh.propose(g), eo.propose(g)):
            assert (x == y).all()
        cols = streams.vote_round_runs(G, members, r, 100, config_id=3)
        if r % 2 == 1:     # lost votes: some groups do not decide, some runs are shorter than others
            keep = rng.random(cols[0].shape[0]) > 0.2
            cols = [np.ascontiguousarray(c[keep]) for c in cols]
        n = cols[0].shape[0]
        d = [torch.zeros(n, dtype=torch.int32, device="cuda") for _ in range(5)] + [torch.zeros(n, dtype=torch.uint8, device="cuda")]
        no, st = torch.zeros(1, dtype=torch.int32, device="cuda"), torch.zeros(n, dtype=torch.uint8, device="cuda")
        dc = [_dev(torch, c) for c in cols]
        eh.profile(2)
        eh.call_dev("accept_reply_batch", n, *[P(c) for c in dc], *[P(t) for t in d], P(no), P(st))
        torch.cuda.synchronize()
        prof = eh.profile_read()
        assert "k_emit_dec_runs" not in prof and "k_merge_runs" not in prof, prof
        assert (list(prof) == ["k_ar_runs_pers"]) if (exchange and n <= 256 * 256) else (sorted(prof) == ["k_ar_runs", "k_runs_check"]), prof
        do = eo.accept_reply(*cols)
        if r % 2 == 0:
            assert int(no.item()) == G
        else:
            assert int(no.item()) < 0
            eh.compact_last_dev()
            torch.cuda.synchronize()
        m = int(no.item())
        got = np.stack([t.cpu().numpy()[:m].astype(np.int32) for t in d], axis=1)
        assert got.shape == do.as_tuple_array().shape and (got == do.as_tuple_array()).all(), f"round {r}"
        assert (st.cpu().numpy() == do.status).all()
    assert_same_state(eh, eo, rng.integers(0, G, 300))
    assert eh.counters() == eo.counters()
    eh.close(), eo.close()


# ---- round 5 -----------------------------------------------------------------------------------------------------------

def test_host_pointer_calls_with_the_lazy_mask_small_irregular_batches(hip_lib, oracle_lib):
    """GPX_LAZY_OUTPUTS set on the engine, HOST-pointer calls of at most 32,768 records (the staged
    one-block path) whose batches are irregular - commits that execute nothing or two slots, ACCEPTs that release
    placeholders, vote runs with lost replies, a broken promise.  gpx.h says the host-pointer calls never leave
    outputs parked: the counts must come back >= 0 and everything must be the oracle's."""
    G = 9000
    eh, eo = make_pair(hip_lib, oracle_lib, 100, G, 3, 8, max_batch=1 << 16)
    mem = np.tile(np.array(NODES, np.int32), (G, 1))
    for e in (eh, eo):
        assert (e.create_groups(np.arange(G), mem, 3, hri_create(G, 3, 100)) == S_OK).all()
    mask = ORDERED_PROPOSE | ORDERED_ACCEPT | ORDERED_COMMIT | ORDERED_REPLY_RUNS
    eh.set_ordered_batches(mask | LAZY_OUTPUTS)
    eo.set_ordered_batches(mask)
    g = np.arange(G, dtype=np.int32)
    z, bc = np.zeros(G, np.int32), np.full(G, 100, np.int32)
    kind = np.full(G, C_HASVALUE, np.uint8)
    rng = np.random.default_rng(9)

    def both(fn, *a):
        ra, rb = getattr(eh, fn)(*a), getattr(eo, fn)(*a)
        return ra, rb

    def same_runs(ra, rb, what):
        (sa, xa), (sb, xb) = ra, rb
        if isinstance(sa, tuple):
            for x, y in zip(sa, sb):
                assert (x == y).all(), what
        else:
            assert (sa == sb).all(), what
        assert xa.as_tuple_array().tolist() == xb.as_tuple_array().tolist(), what
    # slot 3 before slot 2 for a third of the groups: an irregular commit batch (nothing executes there)
    for x, y in zip(*both("propose", g)):
        assert (x == y).all()
    same_runs(*both("accept", g, z, bc, np.ones(G, np.int32), z), "accept slot 1")
    same_runs(*both("commit", g, z, bc, np.ones(G, np.int32), z, kind), "commit slot 1")
    sl = np.where(g % 3 == 0, 3, 2).astype(np.int32)
    same_runs(*both("commit", g, z, bc, sl, z, kind), "commit 3 before 2")
    gg = g[g % 3 == 0]
    n3 = gg.shape[0]
    same_runs(*both("commit", gg, z[:n3], bc[:n3], np.full(n3, 2, np.int32), z[:n3], kind[:n3]), "commit: two slots execute")
    # placeholders (commits without a value), then the ACCEPTs that release them: an irregular ACCEPT batch
    ph = np.zeros(G, np.uint8)
    same_runs(*both("commit", g, z, bc, np.full(G, 4, np.int32), z, ph), "placeholders")
    same_runs(*both("accept", g, z, bc, np.full(G, 4, np.int32), z), "accepts release placeholders")
    # a broken promise in a small batch: applied up to the violation, refused from there on
    gb = g.copy()
    gb[5000] = gb[4999] - 7
    ra, rb = both("accept", gb, z, bc, np.full(G, 5, np.int32), z)
    same_runs(ra, rb, "broken promise")
    assert (ra[0][4][5000:] == S_UNORDERED).all()
    # votes as three runs with lost replies (under the runs promise): irregular, compacted inside the call
    for r in range(2):
        for x, y in zip(*both("propose", g)):
            assert (x == y).all()
        rows = eo.snapshot(g)[0]
        cols = [c.copy() for c in streams.vote_round_runs(G, NODES, 0, 100, config_id=3)]
        cols[3] = (rows["next_proposal_slot"][cols[0]] - 1).astype(np.int32)
        cols[5] = cols[3] - 1
        keep = rng.random(cols[0].shape[0]) > (0.25 if r == 0 else 0.0)
        cols = [np.ascontiguousarray(c[keep]) for c in cols]
        assert cols[0].shape[0] <= 32768
        dh, do = both("accept_reply", *cols)
        assert dh.as_tuple_array().tolist() == do.as_tuple_array().tolist() and (dh.status == do.status).all(), f"votes {r}"
    assert_same_state(eh, eo, rng.integers(0, G, 300))
    assert eh.counters() == eo.counters()
    eh.close(), eo.close()


def test_compact_last_dev_refuses_after_another_call(hip_lib):
    """a stale pending compaction (another batch call came in between) is refused, not launched
    with the earlier call's pointers."""
    import torch
    from gigapaxos_amd._abi import GpxError
    G = 70_000
    e = Engine(hip_lib, 100, G, kmax=3, window=8, max_batch=G + 64)
    mem = np.tile(np.array(NODES, np.int32), (G, 1))
    assert (e.create_groups(np.arange(G), mem, 3, hri_create(G, 3, 100)) == S_OK).all()
    e.set_ordered_batches(ORDERED_PROPOSE | ORDERED_COMMIT | LAZY_OUTPUTS)
    i32 = lambda n: torch.zeros(n, dtype=torch.int32, device="cuda")  # noqa: E731
    u8 = lambda n: torch.zeros(n, dtype=torch.uint8, device="cuda")  # noqa: E731
    P = lambda t: t.data_ptr()  # noqa: E731
    g = torch.arange(G, dtype=torch.int32, device="cuda")
    sl = torch.full((G,), 3, dtype=torch.int32, device="cuda")   # slot 3 first: executes nothing -> parked, count < 0
    cols = [g, i32(G), torch.full((G,), 100, dtype=torch.int32, device="cuda"), sl, i32(G)]
    kind = torch.full((G,), C_HASVALUE, dtype=torch.uint8, device="cuda")
    o = [u8(G)] + [i32(G) for _ in range(3)] + [i32(1)]
    e.call_dev("commit_batch", G, *[P(c) for c in cols], P(kind), *[P(t) for t in o])
    e.sync()
    assert int(o[-1].item()) < 0
    p = [i32(G) for _ in range(4)] + [u8(G)]
    e.call_dev("propose_batch", G, P(g), 0, *[P(t) for t in p])      # another batch call in between
    with pytest.raises(GpxError):
        e.compact_last_dev()
    e.compact_last_dev()                                              # nothing pending any more: a no-op
    e.sync()
    e.close()
