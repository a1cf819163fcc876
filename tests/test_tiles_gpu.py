"""The tiled front end of the accept-reply call (gigapaxos_amd/csrc/gpx_tiles.hip.h: every call that is one pass over
the table with 16-byte aligned columns) against the oracle: every table size and replica count BASELINE names (1 M
groups x 3 and x 5, the 125,000-group shard of config #4's eight-way split), votes that do not fit the 8-byte record
(other ballots, node ids beyond 16 bits, slots and checkpoints far from the reference vote's), groups that are NOT in
lock-step (wide tiles), an odd first vote, streams that are not shuffled, and skewed streams - a bucket with more votes
than the LDS staging holds (copied into its X.rec region first).  Follows PISM.handleBatchedAcceptReply
(PaxosInstanceStateMachine.java:1370-1419)."""
import numpy as np
import pytest

from gigapaxos_amd import hri_create, streams, S_OK
from tests.parity_common import make_pair, assert_same_state
from tests.test_fullsize_gpu import _same, _vote_stream_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _tiles(monkeypatch):
    monkeypatch.setenv("GPX_AR_TILES", "1")


def _took_tiles(eh):
    return "k_scatter_tiles" in eh.profile_read()


def test_partition_front_end_at_the_same_size(hip_lib, oracle_lib, monkeypatch):
    """GPX_AR_TILES=0: k_hist + k_scatter_ar16 for a shape the tiled front end takes by default (what is left to the
    partition front end otherwise: tables beyond 4 M groups - one pass per range -, unaligned columns, calls that
    bring so many votes per bucket that they are split into passes)."""
    monkeypatch.setenv("GPX_AR_TILES", "0")
    _vote_stream_parity(hip_lib, oracle_lib, 1_000_000, 3, True, R=2)


def test_shuffled_votes_under_the_runs_hint_through_tiles(hip_lib, oracle_lib, monkeypatch):
    """GPX_TRY_REPLY_RUNS on every call (the test switch GPX_TRY_RUNS=1): the runs check judges the shuffled batch first,
    the tiled kernels are launched behind its gate word and take the batch."""
    monkeypatch.setenv("GPX_TRY_RUNS", "1")
    _vote_stream_parity(hip_lib, oracle_lib, 1_000_000, 3, True, R=2)


def test_votes_without_a_status_column_through_tiles(hip_lib, oracle_lib):
    """`status` is nullable on the accept-reply call (include/gpx.h): the scatter's prefill and the replay's marks are
    skipped, the decisions are the same."""
    import torch
    G, k = 1_000_000, 3
    members = [100, 101, 102]
    eh, eo = make_pair(hip_lib, oracle_lib, 100, G, k, 8, max_batch=3 * G + 3 * G // 50 + 4096)
    mem = np.tile(np.array(members, np.int32), (G, 1))
    for e in (eh, eo):
        assert (e.create_groups(np.arange(G), mem, k, hri_create(G, k, 100)) == S_OK).all()
    g = np.arange(G, dtype=np.int32)
    for x, y in zip(eh.propose(g), eo.propose(g)):
        assert (x == y).all()
    cols = streams.vote_round(G, members, 0, 100, config_id=3, mix=True)
    n = cols[0].shape[0]
    dc = [torch.from_numpy(c).cuda() for c in cols]
    d = [torch.zeros(n, dtype=torch.int32, device="cuda") for _ in range(5)] + [torch.zeros(n, dtype=torch.uint8, device="cuda")]
    no = torch.zeros(1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    eh.profile(2)
    eh.call_dev("accept_reply_batch", n, *[t.data_ptr() for t in dc], *[t.data_ptr() for t in d], no.data_ptr(), 0)
    eh.sync()
    assert _took_tiles(eh)
    do = eo.accept_reply(*cols)
    m = int(no.item())
    got = np.stack([t[:m].cpu().numpy().astype(np.int32) for t in d], axis=1)
    assert got.shape == do.as_tuple_array().shape and (got == do.as_tuple_array()).all()
    assert eh.snapshot(g)[0].tobytes() == eo.snapshot(g)[0].tobytes()
    eh.close()
    eo.close()


def test_config3_stream_500k_groups_through_tiles(hip_lib, oracle_lib):
    """977 buckets."""
    _vote_stream_parity(hip_lib, oracle_lib, 500_000, 3, True, R=3)


@pytest.mark.parametrize("G,k", [(1_000_000, 5), (125_000, 5), (250_000, 5), (40_000, 3)])
def test_config4_shapes_through_tiles(hip_lib, oracle_lib, G, k):
    """Config #4 on one engine (five replicas at 1 M groups: 306 tiles of 16,384 votes - beyond round 5's 192-workgroup
    gate) and its shards (125,000 / 250,000 groups: a few hundred buckets), a small table."""
    _vote_stream_parity(hip_lib, oracle_lib, G, k, True, R=2)


@pytest.mark.parametrize("G,k", [(60_000, 7), (60_000, 8), (40_000, 12), (30_000, 16)])
def test_larger_groups_through_tiles(hip_lib, oracle_lib, G, k):
    """Groups of 6-8 and 9-16 replicas (PC.MAX_GROUP_SIZE = 16, PaxosConfig.java:532): the KMAX = 8 and KMAX = 16
    instantiations of the per-bucket kernel - eight staging rows per group, so 12 and 16 votes per group send every bucket
    through the general two-pass regrouping - with the adversarial mix (escapes, compacted outputs)."""
    _vote_stream_parity(hip_lib, oracle_lib, G, k, True, R=2)
    # ... and that such a call IS the tiled front end's (the helper above does not look at the kernels)
    members = list(range(100, 100 + k))
    eh, eo = make_pair(hip_lib, oracle_lib, 101, G, k, 8, max_batch=G * k + G * k // 50 + 4096)
    mem = np.tile(np.array(members, np.int32), (G, 1))
    for e in (eh, eo):
        assert (e.create_groups(np.arange(G), mem, k, hri_create(G, k, 101)) == S_OK).all()
    g = np.arange(G, dtype=np.int32)
    for x, y in zip(eh.propose(g), eo.propose(g)):
        assert (x == y).all()
    cols = streams.vote_round(G, members, 0, 101, config_id=4, mix=True)
    eh.profile(2)
    dh, do = eh.accept_reply(*cols), eo.accept_reply(*cols)
    ran = eh.profile_read()
    eh.profile(0)
    assert "k_scatter_tiles" in ran and "k_bucket_ar16_tiles" in ran, sorted(ran)
    _same(dh, do, "profiled call")
    eh.close()
    eo.close()


@pytest.mark.parametrize("T,NT", [(4096, 512), (8192, 512), (8192, 1024), (12288, 1024), (16384, 1024), (4096, 1024)])
def test_every_tile_shape(hip_lib, oracle_lib, monkeypatch, T, NT):
    """The scatter kernel's instantiations (votes and threads per workgroup; chosen per call otherwise), each on a call
    whose last tile is partly filled."""
    monkeypatch.setenv("GPX_TILE_T", str(T))
    monkeypatch.setenv("GPX_TILE_NT", str(NT))
    _vote_stream_parity(hip_lib, oracle_lib, 300_000, 3, True, R=2)


def _lockstep_free_rows(G, k, me, rng):
    """Hot-restore rows of groups that are NOT in lock-step: every group at its own slot (HotRestoreInfo.java:35-157)."""
    rows = hri_create(G, k, me)
    base = rng.integers(1, 2_000_000, G).astype(np.int32)
    rows["acc_slot"] = base
    rows["acc_gc_slot"] = base - 1 - rng.integers(0, 3, G).astype(np.int32)
    rows["next_proposal_slot"] = base
    rows["node_slots"][:, :k] = (base - 1 - rng.integers(0, 400, G).astype(np.int32))[:, None]
    return rows


@pytest.mark.parametrize("odd_first", [False, True])
def test_groups_out_of_lock_step_wide_tiles(hip_lib, oracle_lib, odd_first):
    """Every group at its own slot (what any deployment that is not a benchmark looks like): no vote's slot fits a byte
    next to the reference vote's, every tile goes WIDE - slot and max_cp travel beside the sorted records (A.ext) - and
    nothing is fetched by arrival index.  Checkpoints up to 400 slots behind.  odd_first: the batch's first vote carries
    another ballot and a far slot (round 5 took vote 0 as the reference: every vote would escape)."""
    G, k, R = 600_000, 3, 2
    members = [100, 101, 102]
    rng = np.random.default_rng(11 + odd_first)
    eh, eo = make_pair(hip_lib, oracle_lib, 100, G, k, 8, max_batch=3 * G + 4096)
    mem = np.tile(np.array(members, np.int32), (G, 1))
    rows = _lockstep_free_rows(G, k, 100, rng)
    for e in (eh, eo):
        assert (e.create_groups(np.arange(G), mem, k, rows) == S_OK).all()
    g = np.arange(G, dtype=np.int32)
    eh.profile(2)
    for r in range(R):
        ph, po = eh.propose(g), eo.propose(g)
        for x, y in zip(ph, po):
            assert (x == y).all()
        slot_g = ph[0]
        gidx = np.repeat(g, k)
        acc = np.tile(np.array(members, np.int32), G)
        slot = np.repeat(slot_g, k)
        maxcp = slot - 1 - rng.integers(0, 400, G * k).astype(np.int32)
        order = rng.permutation(G * k)
        cols = [gidx[order], np.zeros(G * k, np.int32), np.full(G * k, 100, np.int32), slot[order], acc[order], maxcp[order]]
        if odd_first:
            cols[1][0] = 1          # a higher ballot in front
            cols[3][0] += 5000
        cols = [np.ascontiguousarray(c) for c in cols]
        dh, do = eh.accept_reply(*cols), eo.accept_reply(*cols)
        _same(dh, do, f"round {r}")
    assert _took_tiles(eh)
    assert eh.snapshot(g)[0].tobytes() == eo.snapshot(g)[0].tobytes()
    assert_same_state(eh, eo, rng.integers(0, G, 48))
    assert eh.counters() == eo.counters()
    eh.close()
    eo.close()


def test_a_few_votes_out_of_lock_step_narrow_tiles(hip_lib, oracle_lib):
    """One group in a hundred at a far slot: fewer than one vote in 32 per tile, the tiles stay narrow and those votes
    fetch slot and max_cp from the caller's columns (ESC_S)."""
    G, k = 300_000, 3
    members = [100, 101, 102]
    rng = np.random.default_rng(3)
    eh, eo = make_pair(hip_lib, oracle_lib, 100, G, k, 8, max_batch=3 * G + 4096)
    mem = np.tile(np.array(members, np.int32), (G, 1))
    rows = hri_create(G, k, 100)
    far = rng.random(G) < 0.01
    rows["acc_slot"][far] = 70_000
    rows["acc_gc_slot"][far] = 69_990
    rows["next_proposal_slot"][far] = 70_000
    for e in (eh, eo):
        assert (e.create_groups(np.arange(G), mem, k, rows) == S_OK).all()
    g = np.arange(G, dtype=np.int32)
    for r in range(2):
        ph, po = eh.propose(g), eo.propose(g)
        for x, y in zip(ph, po):
            assert (x == y).all()
        slot = np.repeat(ph[0], k)
        order = rng.permutation(G * k)
        cols = [np.repeat(g, k)[order], np.zeros(G * k, np.int32), np.full(G * k, 100, np.int32), slot[order],
                np.tile(np.array(members, np.int32), G)[order], (slot - 1)[order]]
        cols = [np.ascontiguousarray(c) for c in cols]
        _same(eh.accept_reply(*cols), eo.accept_reply(*cols), f"round {r}")
    assert eh.snapshot(g)[0].tobytes() == eo.snapshot(g)[0].tobytes()
    eh.close()
    eo.close()


@pytest.mark.parametrize("n_votes", [1025, 4096, 4097, 20_000])
def test_small_calls_through_tiles(hip_lib, oracle_lib, n_votes):
    """Calls just beyond the one-workgroup kernel's 1,024 votes: one or a few tiles, most buckets empty."""
    G, k = 100_000, 3
    members = [100, 101, 102]
    rng = np.random.default_rng(n_votes)
    eh, eo = make_pair(hip_lib, oracle_lib, 100, G, k, 8, max_batch=1 << 16)
    mem = np.tile(np.array(members, np.int32), (G, 1))
    for e in (eh, eo):
        assert (e.create_groups(np.arange(G), mem, k, hri_create(G, k, 100)) == S_OK).all()
    groups = np.sort(rng.choice(G, (n_votes + k - 1) // k, replace=False)).astype(np.int32)
    eh.profile(2)
    for r in range(2):
        for x, y in zip(eh.propose(groups), eo.propose(groups)):
            assert (x == y).all()
        cols = [np.ascontiguousarray(c[:n_votes]) for c in streams.vote_round(G, members, r, 100, config_id=3, groups=groups)]
        _same(eh.accept_reply(*cols), eo.accept_reply(*cols), f"round {r}")
    assert _took_tiles(eh)
    assert eh.snapshot(groups)[0].tobytes() == eo.snapshot(groups)[0].tobytes()
    assert eh.counters() == eo.counters()
    eh.close()
    eo.close()


def test_wide_node_ids_and_ballots_through_tiles(hip_lib, oracle_lib):
    """Entries that escape the 8-byte form re-read their fields from the caller's columns."""
    _vote_stream_parity(hip_lib, oracle_lib, 1_000_000, 3, True, R=3, big_ids=True)


@pytest.mark.parametrize("hot_votes", [3000, 60_000])
def test_skewed_stream_long_runs_and_big_bucket(hip_lib, oracle_lib, hot_votes):
    """`hot_votes` extra votes aimed at the groups of ONE bucket (duplicates of their real votes, shuffled in): long runs
    in every tile, and with 60,000 of them the bucket exceeds the LDS staging as well.  Several slots outstanding per
    group, a slot far from the reference vote's (an escaped entry).  The profile must show that the tiled kernels ran."""
    G, k, R = 1_000_000, 3, 3
    members = [100, 101, 102]
    eh, eo = make_pair(hip_lib, oracle_lib, 100, G, k, 8, max_batch=3 * G + 3 * G // 50 + hot_votes + 4096)
    mem = np.tile(np.array(members, np.int32), (G, 1))
    for e in (eh, eo):
        assert (e.create_groups(np.arange(G), mem, k, hri_create(G, k, 100)) == S_OK).all()
    g = np.arange(G, dtype=np.int32)
    rng = np.random.default_rng(hot_votes)
    eh.profile(2)
    for r in range(R):
        for x, y in zip(eh.propose(g), eo.propose(g)):
            assert (x == y).all()
        cols = [c.copy() for c in streams.vote_round(G, members, r, 100, config_id=3, mix=(r == 1))]
        n0 = cols[0].shape[0]
        hot_groups = 512 * 777 + rng.integers(0, 512, hot_votes)          # one bucket of 512 groups
        pick = rng.integers(0, n0, hot_votes)
        extra = [c[pick].copy() for c in cols]
        extra[0] = hot_groups.astype(np.int32)
        extra[4] = rng.choice(members, hot_votes).astype(np.int32)
        if r == 2:
            extra[3][: hot_votes // 2] += 1000                              # slots a byte cannot reach from vote 0's
        cols = [np.concatenate([c, x]) for c, x in zip(cols, extra)]
        order = rng.permutation(cols[0].shape[0])
        cols = [np.ascontiguousarray(c[order]) for c in cols]
        dh, do = eh.accept_reply(*cols), eo.accept_reply(*cols)
        _same(dh, do, f"round {r}")
    assert _took_tiles(eh)
    assert eh.snapshot(g)[0].tobytes() == eo.snapshot(g)[0].tobytes()
    assert_same_state(eh, eo, np.concatenate([rng.integers(0, G, 40), 512 * 777 + rng.integers(0, 512, 40)]))
    assert eh.counters() == eo.counters()
    eh.close()
    eo.close()


def test_unshuffled_streams_through_tiles(hip_lib, oracle_lib):
    """Streams that are NOT shuffled, sent without a hint: votes sorted by group (a tile of 16,384 votes covers eleven
    buckets: a bucket's records are one or two long runs), the three acceptors' ascending runs with the adversarial mix
    inside (a tile covers 32 buckets), and the sorted stream with the mix.  The tiled kernels must have taken them - not
    the runs kernel."""
    G, k = 1_000_000, 3
    members = [100, 101, 102]
    eh, eo = make_pair(hip_lib, oracle_lib, 100, G, k, 8, max_batch=3 * G + 3 * G // 25 + 4096)
    mem = np.tile(np.array(members, np.int32), (G, 1))
    for e in (eh, eo):
        assert (e.create_groups(np.arange(G), mem, k, hri_create(G, k, 100)) == S_OK).all()
    g = np.arange(G, dtype=np.int32)
    eh.profile(2)
    rounds = [lambda r: streams.vote_round(G, members, r, 100, config_id=3, shuffled=False),
              lambda r: streams.vote_round_runs(G, members, r, 100, config_id=3, mix=True),
              lambda r: streams.vote_round(G, members, r, 100, config_id=3, shuffled=False, mix=True)]
    for r, gen in enumerate(rounds):
        for x, y in zip(eh.propose(g), eo.propose(g)):
            assert (x == y).all()
        cols = gen(r)
        if r == 0:
            assert (np.diff(cols[0]) >= 0).all()
        dh, do = eh.accept_reply(*cols), eo.accept_reply(*cols)
        _same(dh, do, f"round {r}")
    prof = eh.profile_read()
    assert "k_scatter_tiles" in prof and not any(name.startswith("k_ar_runs") for name in prof), prof
    assert eh.snapshot(g)[0].tobytes() == eo.snapshot(g)[0].tobytes()
    assert_same_state(eh, eo, np.random.default_rng(5).integers(0, G, 64))
    assert eh.counters() == eo.counters()
    eh.close()
    eo.close()
