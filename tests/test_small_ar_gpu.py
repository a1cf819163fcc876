"""Small accept-reply calls in ONE launch (gigapaxos_amd/csrc/gpx_small.hip.h: k_ar_small) against the oracle, and the
partition pipeline (GPX_SAR_VOTES_PER_WG=0) on the same streams: uniform and skewed batches, tables of 7 to
1,000,000 groups, hot groups that take several passes, votes outside the table, groups that do not exist, the
sorted-runs hint in front of it."""
import os

import numpy as np
import pytest

from gigapaxos_amd import Engine, hri_create, streams, S_OK, TRY_REPLY_RUNS
from tests.parity_common import assert_same_state, create_mixed_groups, fuzz, make_pair

pytestmark = pytest.mark.gpu


class sar_env:
    """GPX_SAR_VOTES_PER_WG / GPX_SAR_MAX_N are read when an engine is created: votes a workgroup of k_ar_small is sized
    for (0 = the path is off), and the largest call that takes it."""

    def __init__(self, v):
        self.v = v

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in ("GPX_SAR_VOTES_PER_WG", "GPX_SAR_MAX_N")}
        if self.v is None:
            os.environ.pop("GPX_SAR_VOTES_PER_WG", None)
        else:
            os.environ["GPX_SAR_VOTES_PER_WG"] = str(self.v)
        # the engine's crossover to the partition pipeline (32,768 votes by default) lifted to the kernel's own limit:
        # these tests are about the kernel
        os.environ["GPX_SAR_MAX_N"] = "131072"

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _pair(hip_lib, oracle_lib, G, K, max_batch, per_wg=None, window=8):
    with sar_env(per_wg):
        eh = Engine(hip_lib, 100, G, kmax=K, window=window, max_batch=max_batch)
    eo = Engine(oracle_lib, 100, G, kmax=K, window=window, max_batch=max_batch)
    return eh, eo


def _same(dh, do, what):
    assert dh.as_tuple_array().shape == do.as_tuple_array().shape, what
    assert (dh.as_tuple_array() == do.as_tuple_array()).all(), what
    assert (dh.status == do.status).all(), what


@pytest.mark.parametrize("G,K,chunk,per_wg", [
    (7, 3, 5000, None), (7, 3, 5000, 0), (1000, 5, 777, None), (1000, 3, 1, None), (30_000, 3, 65536, None),
    (30_000, 5, 131072, 256), (100_000, 3, 40_000, 64), (100_000, 16, 20_000, None), (1_000_000, 3, 65536, None),
    (1_000_000, 3, 131072, None), (1_000_000, 5, 50_000, 0)])
def test_small_calls_vs_oracle(hip_lib, oracle_lib, G, K, chunk, per_wg):
    """Three outstanding slots per group, their votes (with duplicates, stale and higher ballots) in one shuffled
    stream, fed in calls of `chunk` votes: every call's decisions and statuses, then the state."""
    members = list(range(100, 100 + K))
    eh, eo = _pair(hip_lib, oracle_lib, G, K, max(chunk, G) + 64, per_wg)
    mem = np.tile(np.array(members, np.int32), (G, 1))
    g = np.arange(G, dtype=np.int32)
    for e in (eh, eo):
        assert (e.create_groups(g, mem, K, hri_create(G, K, 100)) == S_OK).all()
    R = 3
    for r in range(R):
        for x, y in zip(eh.propose(g), eo.propose(g)):
            assert (x == y).all()
    rng = np.random.default_rng(G + K)
    live = g if G <= 100_000 else rng.choice(G, 120_000, replace=False).astype(np.int32)
    rounds = [streams.vote_round(G, members, r, 100, mix=True, groups=live) for r in range(R)]
    cols = [np.concatenate([rd[c] for rd in rounds]) for c in range(6)]
    order = rng.permutation(cols[0].shape[0])
    cols = [np.ascontiguousarray(c[order]) for c in cols]
    # a few votes outside the table
    bad = rng.integers(0, cols[0].shape[0], 20)
    cols[0][bad] = rng.choice([-1, G, G + 5, 2 ** 31 - 1, -2 ** 31], 20)
    N = cols[0].shape[0]
    eh.profile(2)
    for o in range(0, N, chunk):
        part = [c[o:o + chunk] for c in cols]
        _same(eh.accept_reply(*part), eo.accept_reply(*part), f"call at {o}")
    prof = eh.profile_read()
    if per_wg == 0:
        assert "k_ar_small" not in prof and "k_bucket_ar16" in prof, prof
    else:
        assert list(prof) == ["k_ar_small"], prof
    assert_same_state(eh, eo, np.unique(np.concatenate([rng.integers(0, G, 300), [0, G - 1]])))
    assert eh.counters() == eo.counters()
    eh.close(), eo.close()


@pytest.mark.parametrize("shape", ["one hot group", "a narrow band", "two bands and a hot group"])
@pytest.mark.parametrize("per_wg", [None, 100])
def test_skewed_small_calls(hip_lib, oracle_lib, shape, per_wg):
    """Batches whose votes crowd into a few groups of a large table: one workgroup's range holds more votes than
    its LDS stages - it narrows the range and, for a single group, walks windows of arrival indices.  The window
    holds eight slots; every slot's votes come many times over (retransmissions)."""
    G, K = 1_000_000, 5
    members = list(range(100, 100 + K))
    eh, eo = _pair(hip_lib, oracle_lib, G, K, 131072 + 64, per_wg)
    rng = np.random.default_rng(len(shape))
    if shape == "one hot group":
        live = np.array([777_777], np.int32)
        n = 30_000
    elif shape == "a narrow band":
        live = np.arange(500_000, 500_040, dtype=np.int32)
        n = 60_000
    else:
        live = np.concatenate([np.arange(10, 30), np.arange(999_000, 999_900, 7), [123_456]]).astype(np.int32)
        n = 131072
    mem = np.tile(np.array(members, np.int32), (live.shape[0], 1))
    for e in (eh, eo):
        assert (e.create_groups(live, mem, K, hri_create(live.shape[0], K, 100)) == S_OK).all()
    for rnd in range(2):
        for r in range(6):
            for x, y in zip(eh.propose(live), eo.propose(live)):
                assert (x == y).all()
        w = np.ones(live.shape[0])
        if shape == "two bands and a hot group":
            w[-1] = 400.0
        gi = rng.choice(live.shape[0], n, p=w / w.sum())
        cols = [live[gi], np.zeros(n, np.int32), np.full(n, 100, np.int32),
                (rng.integers(1, 7, n) + 6 * rnd).astype(np.int32), rng.choice(members + [99], n).astype(np.int32),
                rng.integers(0, 6 * (rnd + 1), n).astype(np.int32)]
        stale = rng.random(n) < 0.01
        cols[2][stale] = 99
        higher = rng.random(n) < 0.0005
        cols[1][higher] = 1
        # groups of the table that were never created, between the live ones
        ghost = rng.random(n) < 0.01
        cols[0][ghost] = (cols[0][ghost] + 1000) % G
        eh.profile(2)
        _same(eh.accept_reply(*cols), eo.accept_reply(*cols), f"{shape} round {rnd}")
        assert list(eh.profile_read()) == ["k_ar_small"]
        assert_same_state(eh, eo, live[:: max(1, live.shape[0] // 50)])
    assert eh.counters() == eo.counters()
    eh.close(), eo.close()


@pytest.mark.parametrize("K,G,seed,per_wg", [(3, 40, 1, None), (5, 3000, 2, 32), (16, 500, 3, None), (3, 200_000, 4, 8)])
def test_small_call_fuzz(hip_lib, oracle_lib, K, G, seed, per_wg):
    """The mixed-operation fuzz of test_parity_gpu.py (colliding slots, duplicate votes, stale and higher ballots,
    non-member acceptors, unknown groups) with every vote batch on the one-launch path, sized for few votes per
    workgroup so that calls of a few hundred votes already spread over many workgroups."""
    with sar_env(per_wg):
        eh, eo = make_pair(hip_lib, oracle_lib, 100, G, K, 8, max_batch=4096)
    rng = np.random.default_rng(seed)
    nodes = list(range(100, 100 + max(K, 5)))
    GL = min(G, 600)  # (create_mixed_groups builds its rows one by one)
    create_mixed_groups(eh, eo, GL, K, nodes, rng)
    fuzz(eh, eo, GL, nodes, rng, steps=120, batch=1500)
    assert_same_state(eh, eo, range(GL))
    assert eh.counters() == eo.counters()
    eh.close(), eo.close()


def test_runs_hint_in_front_of_the_small_path(hip_lib, oracle_lib):
    """GPX_TRY_REPLY_RUNS on small calls: a batch of ascending runs is k_ar_runs<SMALL>'s, a shuffled one is handed
    on to k_ar_small launched behind it (which returns at once for the other)."""
    G, K = 20_000, 3
    members = [100, 101, 102]
    eh, eo = _pair(hip_lib, oracle_lib, G, K, K * G + 4096)
    mem = np.tile(np.array(members, np.int32), (G, 1))
    g = np.arange(G, dtype=np.int32)
    for e in (eh, eo):
        assert (e.create_groups(g, mem, K, hri_create(G, K, 100)) == S_OK).all()
    eh.set_ordered_batches(TRY_REPLY_RUNS)
    for r in range(4):
        for x, y in zip(eh.propose(g), eo.propose(g)):
            assert (x == y).all()
        cols = (streams.vote_round_runs if r % 2 == 0 else streams.vote_round)(G, members, r, 100, mix=True)
        eh.profile(2)
        _same(eh.accept_reply(*cols), eo.accept_reply(*cols), f"round {r}")
        prof = eh.profile_read()
        assert "k_ar_runs_small" in prof and "k_ar_small" in prof and "k_bucket_ar16" not in prof, prof
    assert_same_state(eh, eo, np.random.default_rng(0).integers(0, G, 300))
    assert eh.counters() == eo.counters()
    eh.close(), eo.close()
