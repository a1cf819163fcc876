"""Scenarios for the rows of SURVEY §8(a) that sit either side of the state machine: RequestBatcher
(a14), roundRobinCoordinator (a13), gap detection (§8f-4).  Each runs against any library behind the
ABI (the oracle on CPU, the HIP engine on the GPU box) and returns plain Python data to compare."""
import numpy as np

from gigapaxos_amd import Engine, hri_create, S_OK, S_NOGROUP, C_HASVALUE
from gigapaxos_amd import wire as W


def request_batch_kat(lib):
    """RequestBatcher.dequeueImpl by hand (RequestBatcher.java:163-239): FIFO per group, a head is
    always taken, followers while bytes <= limit and sizes <= MAX_BATCH_SIZE, a refused request
    opens the next batch."""
    e = Engine(lib, 100, 16, kmax=3, window=8)
    we = W.WireEngine(e)
    #       gidx est weight stop
    rows = [(3, 10, 1, 0), (3, 10, 1, 0), (5, 10, 1, 0), (3, 10, 1, 1), (3, 10, 1, 0), (-1, 10, 1, 0),
            (5, 10, 1, 0), (3, 10, 1, 0), (7, 999, 1, 0), (7, 1, 5, 0), (7, 1, 5, 0), (7, 1, 1, 0), (16, 1, 1, 0)]
    a = np.array(rows, np.int32)
    leader, status, b = W.request_batch(we, a[:, 0], a[:, 1], a[:, 2], a[:, 3].astype(np.uint8), max_bytes=25,
                                        max_size=10)
    assert leader.tolist() == [0, 0, 2, 3, 3, -1, 2, 7, 8, 9, 9, 11, -1]
    assert status.tolist() == [0, 0, 0, 0, 0, S_NOGROUP, 0, 0, 0, 0, 0, 0, S_NOGROUP]
    assert b["gidx"].tolist() == [3, 3, 3, 5, 7, 7, 7]
    assert b["leader"].tolist() == [0, 3, 7, 2, 8, 9, 11]
    assert b["count"].tolist() == [2, 2, 1, 2, 1, 2, 1]
    assert b["bytes"].tolist() == [20, 20, 10, 20, 999, 2, 1]
    assert b["size"].tolist() == [2, 2, 1, 2, 1, 10, 1]
    assert b["stop"].tolist() == [0, 1, 0, 0, 0, 0, 0]
    e.close()


def request_batch_run(lib, seed, n=20000, G=3000, hot=True):
    rng = np.random.default_rng(seed)
    e = Engine(lib, 100, G, kmax=3, window=8, max_batch=1 << 16)
    we = W.WireEngine(e)
    g = rng.integers(-1, G + 1, n).astype(np.int32)
    if hot:
        g[rng.random(n) < 0.2] = 17   # one group with thousands of queued requests
        g[rng.random(n) < 0.02] = 18
    est = rng.integers(1, 400, n).astype(np.int32)
    wt = rng.choice([1, 1, 1, 2, 7, 300], size=n).astype(np.int32)
    stop = (rng.random(n) < 0.01).astype(np.uint8)
    out = []
    for kw in (dict(weight=wt, is_stop=stop, max_bytes=2000, max_size=400),
               dict(weight=None, is_stop=None, max_bytes=1 << 20, max_size=2000)):
        leader, status, b = W.request_batch(we, g, est, **kw)
        out.append((leader.tolist(), status.tolist(), {k: v.tolist() for k, v in b.items()}))
    e.close()
    return out


def coordinator_run(lib, seed=0):
    rng = np.random.default_rng(seed)
    G, k = 500, 5
    e = Engine(lib, 100, G, kmax=k, window=8)
    we = W.WireEngine(e)
    ks = rng.integers(1, k + 1, G).astype(np.uint8)
    members = np.zeros((G, k), np.int32)
    for g in range(G):
        members[g, :ks[g]] = np.sort(rng.choice(np.arange(90, 120), size=ks[g], replace=False))
    created = np.arange(G - 20, dtype=np.int32)
    assert (e.create_groups(created, members[created], ks[created], hri_create(G - 20, k, 100)) == S_OK).all()
    names = [b"name-%d-%s" % (g, bytes(rng.integers(0x20, 0x100, int(rng.integers(0, 30))).astype(np.uint8)))
             for g in range(G)]
    named = np.array([g for g in range(G) if g % 9 != 4], np.int32)
    assert (we.bind([names[g] for g in named], named) == S_OK).all()
    q = np.arange(-1, G + 1, dtype=np.int32)
    out = [W.names_coordinator(we, q, b).tolist() for b in (0, 1, 7, -5, 2**31 - 1)]
    # Math.abs(Integer.MIN_VALUE) is negative: the Java indexes members[negative]
    h0 = W.java_string_hash(names[0])
    bal = ((-2**31 - h0 + 2**31) % 2**32) - 2**31
    out.append(W.names_coordinator(we, [0], bal).tolist())
    e.close()
    return out, names, members, ks


def gap_run(lib, seed=0):
    """Commits arriving out of order (decisions with values, meta-commits with and without their
    accept) leave gaps; per group: getMaxCommittedSlot, the missing set, shouldSync."""
    rng = np.random.default_rng(seed)
    G = 400
    e = Engine(lib, 100, G, kmax=3, window=16, max_batch=1 << 14)
    we = W.WireEngine(e)
    mem = np.tile(np.array([100, 101, 102], np.int32), (G - 10, 1))
    assert (e.create_groups(np.arange(G - 10), mem, 3, hri_create(G - 10, 3, 100)) == S_OK).all()
    n = 1500
    g = rng.integers(0, G - 10, n).astype(np.int32)
    slot = rng.integers(1, 14, n).astype(np.int32)
    z = np.zeros(n, np.int32)
    c100 = np.full(n, 100, np.int32)
    # some accepts first (so that a later meta-commit finds its value)
    sel = rng.random(n) < 0.4
    e.accept(g[sel], z[sel], c100[sel], slot[sel], z[sel])
    kind = rng.choice([0, 0, C_HASVALUE, C_HASVALUE | 2], size=n, p=[0.35, 0.35, 0.29, 0.01]).astype(np.uint8)
    e.commit(g, z, c100, slot, z, kind)
    q = np.arange(-1, G + 1, dtype=np.int32)
    out = []
    for thr, mode, lim in ((1, W.SYNC_DEFAULT, 64), (5, W.SYNC_DEFAULT, 4), (400, W.SYNC_TO_PAUSE, 64),
                           (1000, W.SYNC_FORCE, 64)):
        out.append([x.tolist() for x in W.gap_scan(we, q, thr, mode, lim)])
    e.close()
    return out


def election_run(lib, seed=0):
    """Groups of mixed size whose acceptor ballots point at various coordinators (moved there by
    PREPAREs); some groups have a local coordinator at or above that ballot.  The fail-over scan for
    several (down, long dead) combinations."""
    rng = np.random.default_rng(seed)
    G, k = 2000, 5
    e = Engine(lib, 102, G, kmax=k, window=8, max_batch=1 << 14)
    we = W.WireEngine(e)
    ks = rng.integers(1, k + 1, G).astype(np.uint8)
    members = np.zeros((G, k), np.int32)
    for g in range(G):
        others = rng.choice([100, 101, 103, 104, 105, 106], size=ks[g] - 1, replace=False)
        members[g, :ks[g]] = np.sort(np.concatenate([[102], others]))
    coord0 = np.array([members[g, rng.integers(0, ks[g])] for g in range(G)], np.int32)
    created = np.arange(G - 50, dtype=np.int32)
    assert (e.create_groups(created, members[created], ks[created], hri_create(G - 50, k, coord0[created])) == S_OK).all()
    # move some acceptor ballots: higher ballots of member and non-member coordinators
    n = 1200
    g = rng.integers(0, G, n).astype(np.int32)
    bn = rng.integers(1, 4, n).astype(np.int32)
    bc = rng.choice([100, 101, 102, 103, 104, 105, 106, 999], size=n).astype(np.int32)
    e.prepare(g, bn, bc, np.ones(n, np.int32))
    out = []
    q = np.arange(-1, G + 1, dtype=np.int32)
    for down, longdead, force in (((), (), False), ((100,), (), False), ((100, 104), (104,), False),
                                  ((100, 101, 103, 104, 105, 106, 999), (999, 105), False), ((), (), True)):
        out.append([x.tolist() for x in W.election_scan(we, q, down, longdead, force)])
    out.append([x.tolist() for x in W.election_scan(we, None, (101,), (), False)])
    e.close()
    return out
