"""Full-size parity: the HIP engine against the CPU oracle at BASELINE.json's sizes — 1,000,000
groups (3,907 buckets: the partition regime bench.py runs in), the config #3 / #4 / #5 streams, and
4 M groups (buckets of 1024 groups).  Bit-exact: the decided (group, slot, ballot,
medianCheckpointedSlot, kind) stream in order, every per-vote status, every HotRestoreInfo row and
the counters.  Follows PaxosInstanceStateMachine.handleBatchedAcceptReply (PISM:1370-1419) ->
PaxosCoordinatorState.handleAcceptReplyMyBallot (PCS:597-640) as restated in oracle/gpx_oracle.cpp."""
import numpy as np
import pytest

from gigapaxos_amd import Engine, hri_create, streams, S_OK, S_NOGROUP, D_DECISION
from tests.parity_common import make_pair, assert_same_state, churn_run

pytestmark = pytest.mark.gpu


def _same(dh, do, what):
    a, b = dh.as_tuple_array(), do.as_tuple_array()
    assert a.shape == b.shape, f"{what}: {a.shape} vs {b.shape}"
    assert (a == b).all(), f"{what}: first difference at row {int(np.nonzero((a != b).any(1))[0][0])}"
    assert (dh.status == do.status).all(), f"{what}: per-vote status"


def _vote_stream_parity(hip_lib, oracle_lib, G, k, mix, R, big_ids=False):
    members = list(range(100, 100 + k))
    if big_ids:  # node ids outside 16 bits and negative: the 16-byte vote record's escape path
        members = [-7, 100, 70000, 1 << 30, (1 << 31) - 1][:k]
    me = members[1]
    nv = G * k + (G * k // 50 if mix else 0)
    eh, eo = make_pair(hip_lib, oracle_lib, me, G, k, 8, max_batch=nv + 4096)
    mem = np.tile(np.array(members, np.int32), (G, 1))
    for e in (eh, eo):
        assert (e.create_groups(np.arange(G), mem, k, hri_create(G, k, me)) == S_OK).all()
    g = np.arange(G, dtype=np.int32)
    for r in range(R):
        for x, y in zip(eh.propose(g), eo.propose(g)):
            assert (x == y).all()
        cols = streams.vote_round(G, members, r, me, config_id=3 if k == 3 else 4, mix=mix)
        dh, do = eh.accept_reply(*cols), eo.accept_reply(*cols)
        _same(dh, do, f"round {r}")
        if not mix:
            assert dh.gidx.shape[0] == G and (dh.kind == D_DECISION).all()
    sh, so = eh.snapshot(g)[0], eo.snapshot(g)[0]
    assert sh.tobytes() == so.tobytes()
    assert_same_state(eh, eo, np.random.default_rng(G + k).integers(0, G, 48))
    assert eh.counters() == eo.counters()
    eh.close()
    eo.close()


@pytest.mark.parametrize("k,mix", [(3, False), (3, True), (5, True)])
def test_config3_config4_streams_1m_groups_vs_oracle(hip_lib, oracle_lib, k, mix):
    """BASELINE config #3 (K = 3: clean and adversarial mix) and the config #4 stream (K = 5, mix)
    at 1 M groups per engine: 3 M / 5 M shuffled votes per round against the oracle."""
    _vote_stream_parity(hip_lib, oracle_lib, 1_000_000, k, mix, R=3)


def test_config3_stream_4m_groups_vs_oracle(hip_lib, oracle_lib):
    """4 M groups on one engine: buckets of 1024 groups (several partition levels / groups per lane
    depending on the build) - same stream, same oracle."""
    _vote_stream_parity(hip_lib, oracle_lib, 4_000_000, 3, True, R=2)


def test_config3_stream_5m_groups_two_range_passes_vs_oracle(hip_lib, oracle_lib):
    """More than 4 M groups on one engine: the accept-reply call runs one pass per range of 4 M groups
    (4096 buckets of 1024 groups), each pass skipping the other range's votes, outputs chained - the
    decided stream must still be the oracle's, grouped by gidx ascending."""
    _vote_stream_parity(hip_lib, oracle_lib, 5_000_000, 3, True, R=2)


def test_16m_votes_in_one_call_vs_oracle(hip_lib, oracle_lib):
    """A call that brings many rounds of votes at once - six slots outstanding per group, 16,000,000 votes over
    1 M groups in ONE gpx_accept_reply_batch: most groups have 17+ votes in the batch (segments past the
    nibble word: ordered by their own lane, a hot group by the workgroup) and the buckets fill the LDS staging
    to the brim.  Round 2's batch sweep fell off a 13x cliff here; the answers must be the oracle's."""
    G, k, R = 1_000_000, 3, 6
    members = [100, 101, 102]
    eh, eo = make_pair(hip_lib, oracle_lib, 100, G, k, 8, max_batch=16_000_000 + 4096)
    mem = np.tile(np.array(members, np.int32), (G, 1))
    for e in (eh, eo):
        assert (e.create_groups(np.arange(G), mem, k, hri_create(G, k, 100)) == S_OK).all()
    g = np.arange(G, dtype=np.int32)
    for r in range(R):
        for x, y in zip(eh.propose(g), eo.propose(g)):
            assert (x == y).all()
    rounds = [streams.vote_round(G, members, r, 100, config_id=3, mix=(r == 2)) for r in range(R)]
    cols = [np.ascontiguousarray(np.concatenate([rd[c] for rd in rounds])[:16_000_000]) for c in range(6)]
    cols[0][123_456:123_456 + 3000] = 777           # and one hot group: 3000 more votes in a row
    dh, do = eh.accept_reply(*cols), eo.accept_reply(*cols)
    _same(dh, do, "16 M votes")
    assert dh.gidx.shape[0] > 5 * G
    assert eh.snapshot(g)[0].tobytes() == eo.snapshot(g)[0].tobytes()
    assert eh.counters() == eo.counters()
    eh.close()
    eo.close()


def test_vote_stream_wide_node_ids_vs_oracle(hip_lib, oracle_lib):
    """Node ids that do not fit the compact vote record (negative, > 65535, Integer.MAX_VALUE) and
    ballots other than the batch's common one: the record's escape path re-reads the caller's
    columns; decisions must not change."""
    _vote_stream_parity(hip_lib, oracle_lib, 200_000, 5, True, R=3, big_ids=True)


def test_config5_churn_1m_live_groups_vs_oracle(hip_lib, oracle_lib):
    """BASELINE config #5's churn (create / delete mid-run, rows reused, late votes for retired
    groups dropped) with 1 M live groups of 5 replicas on one engine."""
    G_live, R, k = 1_000_000, 3, 5
    cap = G_live + 4 * (G_live // 1000)
    eh, eo = make_pair(hip_lib, oracle_lib, 100, cap, k, 8, max_batch=(G_live + G_live // 500) * k + 4096)
    (oh, oo), _ = churn_run([eh, eo], G_live, cap, R, k, seed=7, churn_frac=0.001)
    for r, (a, b) in enumerate(zip(oh, oo)):
        for x, y, nm in zip(a, b, ("decisions", "vote status", "propose out", "propose status",
                                    "retired rows", "retire status", "create status")):
            if isinstance(x, bytes):
                assert x == y, f"round {r} {nm}"
            else:
                assert x.shape == y.shape and (x == y).all(), f"round {r} {nm}"
        assert a[0].shape[0] == G_live
        if r > 0:
            assert (a[1] == S_NOGROUP).sum() == k * (G_live // 1000)
    sh, so = eh.snapshot(np.arange(cap))[0], eo.snapshot(np.arange(cap))[0]
    assert sh.tobytes() == so.tobytes()
    assert eh.counters() == eo.counters()


def _churn_across_ranges(hip_lib, oracle_lib, G_live, k, R, seed):
    """Config #5 on ONE engine whose table needs several group-range passes per accept-reply call
    (more than 4 M groups: 4096 buckets of 1024 groups per pass).  The free list holds exactly one
    round's worth of rows, so the rows retired in round r are re-created at the end of round r + 1
    and vote again (slot 1) in round r + 2: late votes reach retired rows of EVERY range while
    fresh and re-created rows live in other ranges (the first fresh rows sit at the top of the
    table = the last range; re-created ones wherever the victims were).  Reference behaviour: a
    packet for a killed instance is dropped (PaxosManager.java:1162-1194,
    PaxosInstanceStateMachine.java:441-447); a re-created one starts from its createHRI row
    (HotRestoreInfo.java:145-157)."""
    n_ret = max(1, int(G_live * 0.001))
    cap = G_live + n_ret
    eh, eo = make_pair(hip_lib, oracle_lib, 100, cap, k, 8, max_batch=(G_live + n_ret) * k + 4096)
    (oh, oo), live = churn_run([eh, eo], G_live, cap, R, k, seed=seed, churn_frac=0.001)
    range_of = lambda g: np.asarray(g) >> 22  # noqa: E731  (4 M groups per pass)
    for r, (a, b) in enumerate(zip(oh, oo)):
        for x, y, nm in zip(a, b, ("decisions", "vote status", "propose out", "propose status",
                                    "retired rows", "retire status", "create status")):
            if isinstance(x, bytes):
                assert x == y, f"round {r} {nm}"
            else:
                assert x.shape == y.shape and (x == y).all(), f"round {r} {nm}"
        assert a[0].shape[0] == G_live, f"round {r}: one decision per live group"
        assert (np.diff(a[0][:, 0]) > 0).all(), f"round {r}: decisions grouped by gidx ascending across the passes"
        assert len(np.unique(range_of(a[0][:, 0]))) == (cap + (1 << 22) - 1) >> 22
        if r > 0:
            assert (a[1] == S_NOGROUP).sum() == k * n_ret  # the late votes, every range
        if r >= 2:
            assert (a[0][:, 1] == 1).sum() >= n_ret  # re-created rows decide their slot 1 again
    sh, so = eh.snapshot(np.arange(cap))[0], eo.snapshot(np.arange(cap))[0]
    assert sh.tobytes() == so.tobytes()
    assert eh.counters() == eo.counters()
    eh.close()
    eo.close()


def test_config5_churn_10m_live_groups_three_range_passes_vs_oracle(hip_lib, oracle_lib):
    """BASELINE config #5 at the size it states: 10,000,000 live groups (K = 3) on one engine, three
    group-range passes per accept-reply call, 0.1 % retire / create per round, two rounds."""
    _churn_across_ranges(hip_lib, oracle_lib, 10_000_000, 3, R=2, seed=11)


def test_config5_churn_5m_live_groups_k5_two_range_passes_vs_oracle(hip_lib, oracle_lib):
    """The same with five replicas (config #4's group size): 4,300,000 live groups, two passes, two rounds (rows re-created
    after a retirement vote again in tests/test_fullsize_gpu.py::test_config5_churn_1m_live_groups_vs_oracle, three rounds)."""
    _churn_across_ranges(hip_lib, oracle_lib, 4_300_000, 5, R=2, seed=12)


@pytest.mark.parametrize("order", ["grouped by group", "shuffled"])
def test_full_round_1m_groups_vs_oracle(hip_lib, oracle_lib, order):
    """The whole pipeline on one replica at 1 M groups: propose -> its own ACCEPTs (handleAccept,
    PISM:1080-1166) -> the accept replies of two acceptors (majority of 3) -> the decisions back as
    commits (handleBatchedCommit / extractExecuteAndCheckpoint, PISM:1480-1528, 1619-1701), two
    rounds.  Batches grouped by group take the direct kernels, shuffled ones (plus duplicates and
    records of a ballot that lost) the partition path (k_bucket16<ACCEPT / COMMIT>, 1,954 buckets):
    reply columns, execution runs, decisions, statuses, HotRestoreInfo rows and counters bit for bit."""
    G, k, R = 1_000_000, 3, 2
    rng = np.random.default_rng(5)
    members = [100, 101, 102]
    eh, eo = make_pair(hip_lib, oracle_lib, 100, G, k, 8, max_batch=2 * G + 8192)
    mem = np.tile(np.array(members, np.int32), (G, 1))
    for e in (eh, eo):
        assert (e.create_groups(np.arange(G), mem, k, hri_create(G, k, 100)) == S_OK).all()
    g = np.arange(G, dtype=np.int32)
    for r in range(R):
        ph, po = eh.propose(g), eo.propose(g)
        for x, y in zip(ph, po):
            assert (x == y).all()
        slot, bnum, bcoord, median = ph[0], ph[1], ph[2], ph[3]
        ag, asl, abn, abc, amed = g, slot, bnum, bcoord, median
        if order == "shuffled":  # + 2 % duplicates and 1 % ACCEPTs of a lower ballot (refused: NACK)
            extra = rng.integers(0, G, G // 50)
            low = rng.integers(0, G, G // 100)
            ag = np.concatenate([g, g[extra], g[low]])
            asl = np.concatenate([slot, slot[extra], slot[low]])
            abn = np.concatenate([bnum, bnum[extra], bnum[low] - 1])
            abc = np.concatenate([bcoord, bcoord[extra], bcoord[low]])
            amed = np.concatenate([median, median[extra], median[low]])
            p = rng.permutation(ag.shape[0])
            ag, asl, abn, abc, amed = ag[p], asl[p], abn[p], abc[p], amed[p]
        fl = np.zeros(ag.shape[0], np.uint8)
        (rh, xh), (ro, xo) = eh.accept(ag, abn, abc, asl, amed, fl), eo.accept(ag, abn, abc, asl, amed, fl)
        for x, y, nm in zip(rh, ro, ("r_bnum", "r_bcoord", "r_maxcp", "r_flags", "status")):
            assert (x == y).all(), f"round {r} accept {nm}"
        assert (xh.as_tuple_array() == xo.as_tuple_array()).all()
        # votes of acceptors 100 and 101, interleaved at random
        vg = np.concatenate([g, g])
        vs = np.concatenate([slot, slot])
        va = np.concatenate([np.full(G, 100, np.int32), np.full(G, 101, np.int32)])
        p = rng.permutation(2 * G)
        vg, vs, va = vg[p], vs[p], va[p]
        vb, vc = np.concatenate([bnum, bnum])[p], np.concatenate([bcoord, bcoord])[p]
        mcp = np.full(2 * G, r, np.int32)
        dh, do = eh.accept_reply(vg, vb, vc, vs, va, mcp), eo.accept_reply(vg, vb, vc, vs, va, mcp)
        _same(dh, do, f"round {r} decisions")
        assert dh.gidx.shape[0] == G
        cg, cs, cb, cc, cm = dh.gidx, dh.slot, dh.bnum, dh.bcoord, dh.median_cp
        ck = np.zeros(G, np.uint8)
        if order == "shuffled":
            p = rng.permutation(G)
            cg, cs, cb, cc, cm = cg[p], cs[p], cb[p], cc[p], cm[p]
        (sh, ch), (so, co) = eh.commit(cg, cb, cc, cs, cm, ck), eo.commit(cg, cb, cc, cs, cm, ck)
        assert (sh == so).all(), f"round {r} commit status"
        assert (ch.as_tuple_array() == co.as_tuple_array()).all(), f"round {r} execution runs"
        assert ch.as_tuple_array().shape[0] == G
    sh, so = eh.snapshot(g)[0], eo.snapshot(g)[0]
    assert sh.tobytes() == so.tobytes()
    assert eh.counters() == eo.counters()
    eh.close()
    eo.close()
