"""SURVEY.md 8(d)'s stream generator (gigapaxos_amd/native/gpx_streams.c: xorshift64* seeded 0x9E3779B97F4A7C15 ^
(config << 32) ^ round, a Fisher-Yates per group for the acceptor order, one Fisher-Yates over the round's records) against
its pure-Python reading (streams.survey_reference), known xorshift64* outputs, and the properties every leg of bench.py
relies on (each group K votes, one per member; the mix's counts; determinism).  CPU only."""
import numpy as np
import pytest

from gigapaxos_amd import streams


@pytest.fixture(scope="module", autouse=True)
def _built():
    if streams.native_streams() is None:
        import __graft_entry__
        __graft_entry__.build()
    assert streams.native_streams() is not None


def test_xorshift64star_first_outputs():
    """The generator itself, from the published recurrence (Vigna, "An experimental exploration of Marsaglia's xorshift
    generators, scrambled", 2016: shifts 12, 25, 27, multiplier 0x2545F4914F6CDD1D), state 1."""
    s, M, out = 1, (1 << 64) - 1, []
    for _ in range(3):
        s ^= s >> 12
        s = (s ^ (s << 25)) & M
        s ^= s >> 27
        out.append((s * 0x2545F4914F6CDD1D) & M)
    assert out[0] == 0x47E4CE4B896CDD1D  # 33554433 * 0x2545F4914F6CDD1D mod 2^64
    assert len(set(out)) == 3


@pytest.mark.parametrize("G,K,cfg,rnd,shuffled,mix", [
    (50, 3, 3, 0, True, False), (200, 5, 4, 7, True, True), (37, 3, 3, 2, False, True), (64, 3, 19, 1, False, False),
    (1, 3, 3, 0, True, True), (300, 1, 3, 5, True, True)])
def test_native_generator_matches_the_python_reading(G, K, cfg, rnd, shuffled, mix):
    members = list(range(100, 100 + K))
    a = streams.vote_round_survey(G, members, rnd, 100, config_id=cfg, shuffled=shuffled, mix=mix)
    b = streams.survey_reference(G, members, rnd, 100, config_id=cfg, shuffled=shuffled, mix=mix)
    if not shuffled and mix:
        o = np.argsort(b[0], kind="stable")
        b = tuple(c[o] for c in b)
    for x, y in zip(a, b):
        assert x.dtype == np.int32 and x.shape == y.shape and (x == y).all()


def test_round_shape_and_determinism():
    G, K = 20_000, 3
    members = [100, 101, 102]
    a = streams.vote_round_survey(G, members, 4, 100, config_id=3)
    b = streams.vote_round_survey(G, members, 4, 100, config_id=3)
    c = streams.vote_round_survey(G, members, 5, 100, config_id=3)
    assert all((x == y).all() for x, y in zip(a, b))
    assert not (a[0] == c[0]).all()
    gidx, bnum, bcoord, slot, acc, maxcp = a
    assert gidx.shape[0] == G * K and (np.bincount(gidx, minlength=G) == K).all()
    assert (bnum == 0).all() and (bcoord == 100).all() and (slot == 5).all() and (maxcp == 4).all()
    # every group hears from every member exactly once
    key = np.sort(gidx.astype(np.int64) * 1000 + acc)
    assert (np.diff(key) > 0).all() and set(np.unique(acc)) == set(members)
    # the shuffle moved things: not sorted, and both halves of the group space appear in the first tenth
    assert (np.diff(gidx) < 0).any() and gidx[: G * K // 10].min() < G // 2 < gidx[: G * K // 10].max()
    # the three acceptor orders of a group are about equally often first (a per-group Fisher-Yates, not a rotation)
    s = streams.vote_round_survey(G, members, 4, 100, config_id=3, shuffled=False)
    first = np.bincount(s[4][::K] - 100, minlength=K) / G
    assert (abs(first - 1 / K) < 0.02).all()


def test_mix_counts():
    G, K = 10_000, 5
    members = list(range(100, 105))
    cols = streams.vote_round_survey(G, members, 0, 100, config_id=4, mix=True)
    n = G * K
    assert cols[0].shape[0] == n + n // 100 + n // 200 + n // 1000
    assert int((cols[2] == 99).sum()) == n // 200 and int((cols[1] == 1).sum()) == n // 1000


def test_empty_round():
    for mix in (False, True):
        cols = streams.vote_round_survey(0, [100, 101, 102], 0, 100, mix=mix)
        assert all(c.shape == (0,) for c in cols)


def test_churn_groups_column():
    """`groups`: the round covers these group indices instead of 0..G-1 (the churn configuration's live set)."""
    live = np.arange(5, 5000, 7, dtype=np.int32)
    cols = streams.vote_round_survey(0, [100, 101, 102], 3, 100, config_id=5, groups=live)
    assert cols[0].shape[0] == 3 * live.shape[0] and set(np.unique(cols[0])) == set(live.tolist())


def test_engine_and_oracle_consume_the_stream(oracle_lib):
    """The stream drives rounds to their decisions on the oracle as the numpy one does: G decisions of slot r + 1; the
    groups that saw a higher ballot in the round with the adversarial mix have lost their coordinator in the next."""
    from gigapaxos_amd import Engine, hri_create, S_OK
    G, K = 3000, 3
    members = [100, 101, 102]
    eo = Engine(oracle_lib, 100, G, kmax=K, window=8)
    assert (eo.create_groups(np.arange(G), np.tile(np.array(members, np.int32), (G, 1)), K, hri_create(G, K, 100)) == S_OK).all()
    lost = 0
    for r in range(3):
        eo.propose(np.arange(G, dtype=np.int32))
        cols = streams.vote_round_survey(G, members, r, 100, config_id=3, mix=(r == 1))
        t = eo.accept_reply(*cols).as_tuple_array()
        assert (t[:, 1] == r + 1).all() and np.unique(t[:, 0]).shape[0] == t.shape[0] == G - lost
        lost += np.unique(cols[0][cols[1] == 1]).shape[0]
    assert lost > 0
    eo.close()
