"""Tiny accept-reply calls in ONE launch of one workgroup (gigapaxos_amd/csrc/gpx_small.hip.h: k_ar_tiny, at most
1,024 votes in any order) against the oracle, and the partition pipeline (GPX_SAR_MAX_N=0) on the same streams:
uniform and skewed batches, tables of 7 to 1,000,000 groups, hot groups (a lane's own replay, the workgroup's
sort), votes outside the table, groups that do not exist, the sorted-runs hint."""
import os

import numpy as np
import pytest

from gigapaxos_amd import Engine, hri_create, streams, S_OK, TRY_REPLY_RUNS
from tests.parity_common import assert_same_state, create_mixed_groups, fuzz, make_pair

pytestmark = pytest.mark.gpu


class sar_env:
    """GPX_SAR_MAX_N is read when an engine is created: the largest call that takes k_ar_tiny (0 = none does)."""

    def __init__(self, tiny):
        self.tiny = tiny

    def __enter__(self):
        self.old = os.environ.get("GPX_SAR_MAX_N")
        if self.tiny:
            os.environ.pop("GPX_SAR_MAX_N", None)
        else:
            os.environ["GPX_SAR_MAX_N"] = "0"

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("GPX_SAR_MAX_N", None)
        else:
            os.environ["GPX_SAR_MAX_N"] = self.old


def _pair(hip_lib, oracle_lib, G, K, max_batch, tiny=True, window=8):
    with sar_env(tiny):
        eh = Engine(hip_lib, 100, G, kmax=K, window=window, max_batch=max_batch)
    eo = Engine(oracle_lib, 100, G, kmax=K, window=window, max_batch=max_batch)
    return eh, eo


def _same(dh, do, what):
    assert dh.as_tuple_array().shape == do.as_tuple_array().shape, what
    assert (dh.as_tuple_array() == do.as_tuple_array()).all(), what
    assert (dh.status == do.status).all(), what


@pytest.mark.parametrize("G,K,chunk,tiny", [
    (7, 3, 1024, True), (7, 3, 1024, False), (1000, 5, 777, True), (1000, 3, 1, True), (30_000, 3, 1024, True),
    (30_000, 5, 900, True), (100_000, 3, 40_000, True), (100_000, 16, 1000, True), (1_000_000, 3, 1024, True),
    (1_000_000, 5, 512, True), (1_000_000, 5, 1024, False)])
def test_small_calls_vs_oracle(hip_lib, oracle_lib, G, K, chunk, tiny):
    """Three outstanding slots per group, their votes (with duplicates, stale and higher ballots) in one shuffled
    stream, fed in calls of `chunk` votes: every call's decisions and statuses, then the state."""
    members = list(range(100, 100 + K))
    eh, eo = _pair(hip_lib, oracle_lib, G, K, max(chunk, G) + 64, tiny)
    mem = np.tile(np.array(members, np.int32), (G, 1))
    g = np.arange(G, dtype=np.int32)
    for e in (eh, eo):
        assert (e.create_groups(g, mem, K, hri_create(G, K, 100)) == S_OK).all()
    R = 3
    for r in range(R):
        for x, y in zip(eh.propose(g), eo.propose(g)):
            assert (x == y).all()
    rng = np.random.default_rng(G + K)
    live = g if G <= 1000 else rng.choice(G, 12_000, replace=False).astype(np.int32)
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
    if not tiny or chunk > 1024:
        assert any(k.startswith("k_bucket_ar16") for k in prof) and (tiny or "k_ar_tiny" not in prof), prof
    else:
        assert list(prof) == ["k_ar_tiny"], prof
    assert_same_state(eh, eo, np.unique(np.concatenate([rng.integers(0, G, 300), [0, G - 1]])))
    assert eh.counters() == eo.counters()
    eh.close(), eo.close()


@pytest.mark.parametrize("shape", ["one hot group", "a narrow band", "two bands and a hot group"])
@pytest.mark.parametrize("n", [1024, 60_000])
def test_skewed_small_calls(hip_lib, oracle_lib, shape, n):
    """Batches whose votes crowd into a few groups of a large table: a lane of k_ar_tiny with more than 16 votes
    replays them itself (ranked by the lane up to 96, by the workgroup's sort beyond: 1,024 votes of ONE group is the
    extreme), the pipeline regroups a bucket in global memory.  The window holds eight slots; every slot's votes
    come many times over (retransmissions)."""
    G, K = 1_000_000, 5
    members = list(range(100, 100 + K))
    eh, eo = _pair(hip_lib, oracle_lib, G, K, 60_000 + 64)
    rng = np.random.default_rng(len(shape))
    if shape == "one hot group":
        live = np.array([777_777], np.int32)
    elif shape == "a narrow band":
        live = np.arange(500_000, 500_040, dtype=np.int32)
    else:
        live = np.concatenate([np.arange(10, 30), np.arange(999_000, 999_900, 7), [123_456]]).astype(np.int32)
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
        assert (list(eh.profile_read()) == ["k_ar_tiny"]) == (n <= 1024)
        assert_same_state(eh, eo, live[:: max(1, live.shape[0] // 50)])
    assert eh.counters() == eo.counters()
    eh.close(), eo.close()


@pytest.mark.parametrize("K,G,seed", [(3, 40, 1), (5, 3000, 2), (16, 500, 3), (3, 200_000, 4)])
def test_small_call_fuzz(hip_lib, oracle_lib, K, G, seed):
    """The mixed-operation fuzz of test_parity_gpu.py (colliding slots, duplicate votes, stale and higher ballots,
    non-member acceptors, unknown groups): every vote batch has at most 1,000 votes - the one-launch path."""
    with sar_env(True):
        eh, eo = make_pair(hip_lib, oracle_lib, 100, G, K, 8, max_batch=4096)
    rng = np.random.default_rng(seed)
    nodes = list(range(100, 100 + max(K, 5)))
    GL = min(G, 600)  # (create_mixed_groups builds its rows one by one)
    create_mixed_groups(eh, eo, GL, K, nodes, rng)
    eh.profile(2)
    fuzz(eh, eo, GL, nodes, rng, steps=120, batch=1000)
    prof = eh.profile_read()
    assert "k_ar_tiny" in prof and not any(k.startswith("k_bucket_ar16") for k in prof), prof
    assert_same_state(eh, eo, range(GL))
    assert eh.counters() == eo.counters()
    eh.close(), eo.close()


@pytest.mark.parametrize("G", [300, 20_000])
def test_runs_hint_and_small_calls(hip_lib, oracle_lib, G):
    """GPX_TRY_REPLY_RUNS: a call of at most 1,024 votes goes straight to k_ar_tiny, in runs or not (the results are
    the same on every path); a larger one is k_ar_runs<SMALL>'s if it is a few ascending runs, else handed on to the
    partition pipeline launched behind it."""
    K = 3
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
        if G == 300:
            assert list(prof) == ["k_ar_tiny"], prof
        else:
            assert "k_ar_runs_pers" in prof and "k_ar_tiny" not in prof, prof
    assert_same_state(eh, eo, np.random.default_rng(0).integers(0, G, 300))
    assert eh.counters() == eo.counters()
    eh.close(), eo.close()
