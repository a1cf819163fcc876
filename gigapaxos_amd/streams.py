"""Deterministic synthetic accept-reply streams (BASELINE configs 3-5).

For round r every group g has slot s = r + 1 outstanding at the coordinator (engine id
`coordinator`) and receives K votes (gidx, bnum=0, bcoord=coordinator, slot=s,
acceptor=members[pi(j)], max_cp=s-1), pi a per-(g, r) permutation; the round's K*G records are
shuffled globally (variant "shuffled") or kept sorted by gidx (variant "sorted").

Adversarial mix (variant mix=True): 1 % duplicated votes, 0.5 % stale-ballot votes
(bcoord - 1), 0.1 % higher-ballot votes (bnum = 1), appended and shuffled in.

Two generators.  `vote_round` (rounds 1-5: every test, every golden fixture): numpy PCG64 seeded with
SEED ^ (config_id << 32) ^ round - a scalar generator cannot fill 3 M-record rounds fast enough in Python, and parity
only needs the oracle and the engine to consume IDENTICAL arrays.  `vote_round_survey` (round 5; what `bench.py` times
by default): SURVEY.md 8(d) to the letter - xorshift64* seeded the same way, a Fisher-Yates per group for the acceptor
order, one Fisher-Yates over the round's records - in C (`gigapaxos_amd/native/gpx_streams.c`, where everything the
survey leaves open is written down), so that the stream can be regenerated from the text alone; `survey_reference` below
is the same thing in pure Python for small cases (the test that pins the C against it).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

SEED = 0x9E3779B97F4A7C15
_NATIVE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "libgpx_streams.so")
_native = None


def native_streams():
    """gigapaxos_amd/native/libgpx_streams.so (built by __graft_entry__.build() with gcc), or None."""
    global _native
    if _native is None and os.path.exists(_NATIVE):
        lib = C.CDLL(_NATIVE)
        lib.gpx_stream_capacity.restype = C.c_int64
        lib.gpx_stream_capacity.argtypes = [C.c_int64, C.c_int32, C.c_int32]
        lib.gpx_stream_vote_round.restype = C.c_int64
        lib.gpx_stream_vote_round.argtypes = [C.c_int64, C.c_void_p, C.c_int32, C.c_void_p] + [C.c_int32] * 5 + [C.c_void_p] * 6
        _native = lib
    return _native


def vote_round_survey(num_groups: int, members, rnd: int, coordinator: int, config_id: int = 3, shuffled: bool = True,
                      mix: bool = False, groups=None):
    """SURVEY.md 8(d)'s stream as specified (xorshift64*, Fisher-Yates): six int32 columns like vote_round."""
    lib = native_streams()
    if lib is None:
        raise RuntimeError("gigapaxos_amd/native/libgpx_streams.so is missing: python -c 'import __graft_entry__ as g; g.build()'")
    members = np.ascontiguousarray(members, np.int32)
    g = None if groups is None else np.ascontiguousarray(groups, np.int32)
    G = num_groups if g is None else g.shape[0]
    cap = int(lib.gpx_stream_capacity(G, members.shape[0], 1 if mix else 0))
    cols = [np.empty(max(cap, 1), np.int32) for _ in range(6)]
    n = int(lib.gpx_stream_vote_round(G, None if g is None else g.ctypes.data, members.shape[0], members.ctypes.data, rnd,
                                      coordinator, config_id, 1 if shuffled else 0, 1 if mix else 0,
                                      *[c.ctypes.data for c in cols]))
    assert n == cap
    cols = [c[:n] for c in cols]
    if not shuffled and mix:  # variant B: sorted by group (the extra votes fall in behind their group's)
        order = np.argsort(cols[0], kind="stable")
        cols = [np.ascontiguousarray(c[order]) for c in cols]
    return tuple(cols)


def survey_reference(num_groups: int, members, rnd: int, coordinator: int, config_id: int = 3, shuffled: bool = True,
                     mix: bool = False):
    """The same stream in pure Python (small cases only): the reading gpx_streams.c is checked against."""
    M = (1 << 64) - 1
    s = (SEED ^ ((config_id & 0xFFFFFFFF) << 32) ^ (rnd & 0xFFFFFFFF)) & M or SEED

    def below(m):
        nonlocal s
        s ^= s >> 12
        s = (s ^ (s << 25)) & M
        s ^= s >> 27
        return (((s * 0x2545F4914F6CDD1D) & M) * m) >> 64
    k = len(members)
    rows = []
    for g in range(num_groups):
        pi = list(range(k))
        for j in range(k - 1, 0, -1):
            t = below(j + 1)
            pi[j], pi[t] = pi[t], pi[j]
        rows += [[g, 0, coordinator, rnd + 1, int(members[pi[j]]), rnd] for j in range(k)]
    n = len(rows)
    if mix and n:
        nd, ns, nh = max(1, n // 100), max(1, n // 200), max(1, n // 1000)
        for q in range(nd + ns + nh):
            r = list(rows[below(n)])
            if nd <= q < nd + ns:
                r[2] -= 1
            if q >= nd + ns:
                r[1] = 1
            rows.append(r)
    if shuffled:
        for i in range(len(rows) - 1, 0, -1):
            t = below(i + 1)
            rows[i], rows[t] = rows[t], rows[i]
    a = np.array(rows, np.int32).reshape(-1, 6)
    return tuple(np.ascontiguousarray(a[:, c]) for c in range(6))


def _rng(config_id: int, rnd: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64((SEED ^ (config_id << 32) ^ rnd) & ((1 << 64) - 1)))


def vote_round(num_groups: int, members, rnd: int, coordinator: int, config_id: int = 3,
               shuffled: bool = True, mix: bool = False, groups=None):
    """One round of votes as six int32 columns (gidx, bnum, bcoord, slot, acceptor, max_cp)."""
    members = np.asarray(members, np.int32)
    k = members.shape[0]
    rng = _rng(config_id, rnd)
    g = np.arange(num_groups, dtype=np.int32) if groups is None else np.asarray(groups, np.int32)
    G = g.shape[0]
    s = rnd + 1
    # per-(g, r) acceptor permutation: argsort of random keys
    perm = np.argsort(rng.random((G, k)), axis=1).astype(np.int32)
    gidx = np.repeat(g, k)
    acceptor = members[perm].reshape(-1)
    n = G * k
    bnum = np.zeros(n, np.int32)
    bcoord = np.full(n, coordinator, np.int32)
    slot = np.full(n, s, np.int32)
    max_cp = np.full(n, s - 1, np.int32)
    cols = [gidx, bnum, bcoord, slot, acceptor, max_cp]
    if mix:
        nd, ns_, nh = max(1, n // 100), max(1, n // 200), max(1, n // 1000)
        pick = rng.integers(0, n, nd + ns_ + nh)
        extra = [c[pick].copy() for c in cols]
        extra[2][nd:nd + ns_] -= 1  # stale: bcoord - 1
        extra[1][nd + ns_:] = 1  # higher ballot: bnum = 1
        cols = [np.concatenate([c, e]) for c, e in zip(cols, extra)]
        n = cols[0].shape[0]
    if shuffled:
        order = rng.permutation(n)
        cols = [np.ascontiguousarray(c[order]) for c in cols]
    elif mix:
        order = np.argsort(cols[0], kind="stable")
        cols = [np.ascontiguousarray(c[order]) for c in cols]
    return tuple(cols)


def fmix32(h):
    """murmur3 finaliser — the group -> GPU shard hash."""
    h = np.asarray(h, np.uint32).copy()
    h ^= h >> np.uint32(16)
    h *= np.uint32(0x85EBCA6B)
    h ^= h >> np.uint32(13)
    h *= np.uint32(0xC2B2AE35)
    h ^= h >> np.uint32(16)
    return h


def shard_of(gidx, n_shards: int):
    return (fmix32(gidx) % np.uint32(n_shards)).astype(np.int32)


def vote_round_runs(num_groups: int, members, rnd: int, coordinator: int, config_id: int = 3, mix: bool = False,
                    groups=None, drop: float = 0.0):
    """The same round as the coordinator of a real cluster sees it: the replies of acceptor 0, then those of
    acceptor 1, ... - every acceptor's replies grouped by group, groups ascending (the order they leave
    gpx_accept_batch in), i.e. len(members) ascending runs.  mix: 1 % duplicated votes (adjacent to the
    original, as a retransmitted frame would sit), 0.5 % stale-ballot and 0.1 % higher-ballot votes inside
    the runs; drop: fraction of every acceptor's replies that is missing (lost frames: the runs differ)."""
    members = np.asarray(members, np.int32)
    k = members.shape[0]
    rng = _rng(config_id ^ 0x5A, rnd)
    g = np.arange(num_groups, dtype=np.int32) if groups is None else np.sort(np.asarray(groups, np.int32))
    s = rnd + 1
    cols = [[] for _ in range(6)]
    for j in range(k):
        gj = g
        if drop > 0.0:
            gj = g[rng.random(g.shape[0]) >= drop]
        n = gj.shape[0]
        bnum = np.zeros(n, np.int32)
        bcoord = np.full(n, coordinator, np.int32)
        if mix:
            dup = rng.random(n) < 0.01
            rep = np.ones(n, np.int64)
            rep[dup] = 2
            gj = np.repeat(gj, rep)
            n = gj.shape[0]
            bnum = np.zeros(n, np.int32)
            bcoord = np.full(n, coordinator, np.int32)
            stale = rng.random(n) < 0.005
            bcoord[stale] -= 1
            higher = rng.random(n) < 0.001
            bnum[higher] = 1
        for c, v in zip(cols, (gj, bnum, bcoord, np.full(n, s, np.int32), np.full(n, members[j], np.int32),
                               np.full(n, s - 1, np.int32))):
            c.append(v)
    return tuple(np.ascontiguousarray(np.concatenate(c)) for c in cols)
