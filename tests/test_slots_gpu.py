"""The slotted front end of the shuffled accept-reply call (gigapaxos_amd/csrc/gpx_slots.hip.h: calls of one partition
pass over at least 820 buckets - tables from about 420,000 groups - and at most 192 scatter workgroups, i.e. 3.1 M votes
at 1 M groups) against the oracle: votes that do not fit the 8-byte slot entry (other ballots, node
ids beyond 16 bits, slots and checkpoints far from vote 0's), and skewed streams - a bucket that gets more votes from one
scatter workgroup than its slot holds (the overflow list) and more votes altogether than the LDS staging holds (copied
into its X.rec region first)."""
import numpy as np
import pytest

from gigapaxos_amd import hri_create, streams, S_OK
from tests.parity_common import make_pair, assert_same_state
from tests.test_fullsize_gpu import _same, _vote_stream_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _slots(monkeypatch):
    monkeypatch.setenv("GPX_AR_SLOTS", "1")


def _took_slots(eh):
    return "k_scatter_slots" in eh.profile_read()


def test_partition_front_end_at_the_same_size(hip_lib, oracle_lib, monkeypatch):
    """GPX_AR_SLOTS=0: k_hist + k_scatter_ar16 for the shape the slotted front end takes by default (tests/
    test_fullsize_gpu.py's K = 3 cases at 1 M groups go through the slots now; K = 5 at 1 M groups, 4 M and 5 M
    groups and every smaller table still take the partition front end)."""
    monkeypatch.setenv("GPX_AR_SLOTS", "0")
    _vote_stream_parity(hip_lib, oracle_lib, 1_000_000, 3, True, R=2)


def test_shuffled_votes_under_the_runs_hint_through_slots(hip_lib, oracle_lib, monkeypatch):
    """GPX_TRY_REPLY_RUNS on every call (the test switch GPX_TRY_RUNS=1): the runs check judges the shuffled batch first,
    the slotted kernels are launched behind its gate word and take the batch."""
    monkeypatch.setenv("GPX_TRY_RUNS", "1")
    _vote_stream_parity(hip_lib, oracle_lib, 1_000_000, 3, True, R=2)


def test_votes_without_a_status_column_through_slots(hip_lib, oracle_lib):
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
    assert _took_slots(eh)
    do = eo.accept_reply(*cols)
    m = int(no.item())
    got = np.stack([t[:m].cpu().numpy().astype(np.int32) for t in d], axis=1)
    assert got.shape == do.as_tuple_array().shape and (got == do.as_tuple_array()).all()
    assert eh.snapshot(g)[0].tobytes() == eo.snapshot(g)[0].tobytes()
    eh.close()
    eo.close()


def test_config3_stream_500k_groups_through_slots(hip_lib, oracle_lib):
    """977 buckets: tiles of 8,192 votes."""
    _vote_stream_parity(hip_lib, oracle_lib, 500_000, 3, True, R=3)


def test_wide_node_ids_and_ballots_through_slots(hip_lib, oracle_lib):
    """Entries that escape the 8-byte form re-read their fields from the caller's columns."""
    _vote_stream_parity(hip_lib, oracle_lib, 1_000_000, 3, True, R=3, big_ids=True)


@pytest.mark.parametrize("hot_votes", [3000, 60_000])
def test_skewed_stream_overflow_and_big_bucket(hip_lib, oracle_lib, hot_votes):
    """`hot_votes` extra votes aimed at the groups of ONE bucket (duplicates of their real votes, shuffled in): its slots
    overflow - more than 24 votes from one 16,384-vote tile - and with 60,000 of them the bucket exceeds the LDS staging
    as well.  Several slots outstanding per group, a slot far from vote 0's (an escaped entry).  The profile must show
    that the slotted kernels ran."""
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
    assert _took_slots(eh)
    assert eh.snapshot(g)[0].tobytes() == eo.snapshot(g)[0].tobytes()
    assert_same_state(eh, eo, np.concatenate([rng.integers(0, G, 40), 512 * 777 + rng.integers(0, 512, 40)]))
    assert eh.counters() == eo.counters()
    eh.close()
    eo.close()


def test_unshuffled_streams_through_the_overflow_segments(hip_lib, oracle_lib):
    """Streams that are NOT shuffled, sent without a hint: votes sorted by group (a tile of 16,384 votes covers eleven
    buckets, nearly every vote leaves its slot for the workgroup's overflow segment, every bucket finds its records
    there by binary search), the three acceptors' ascending runs with the adversarial mix inside (a tile covers 32
    buckets), and the sorted stream with the mix.  The slotted kernels must have taken them - not the runs kernel."""
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
    assert "k_scatter_slots" in prof and not any(name.startswith("k_ar_runs") for name in prof), prof
    assert eh.snapshot(g)[0].tobytes() == eo.snapshot(g)[0].tobytes()
    assert_same_state(eh, eo, np.random.default_rng(5).integers(0, G, 64))
    assert eh.counters() == eo.counters()
    eh.close()
    eo.close()
