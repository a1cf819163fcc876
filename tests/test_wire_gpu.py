"""Wire codec parity: the HIP kernels (through the C-ABI of include/gpx_wire.h) against the CPU
oracle on identical frame bursts — every per-frame status / row / type, every decoded column, and
the packed BATCHED_COMMIT bytes, bit for bit."""
import numpy as np
import pytest

from gigapaxos_amd import Engine, hri_create, streams, S_OK, D_DECISION
from gigapaxos_amd import wire as W
from tests import test_wire_oracle as scen
from tests.wire_common import make_wire_pair, random_frames, assert_same_decode

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["512-frame tiles", "256-frame tiles"])
def _decode_path(request, monkeypatch):
    """gpx_wire_decode is ONE launch (k_wire_decode1: look-back over tiles of 512 frames, or of 256 with
    GPX_WD_TILE=256, read when the engine first touches the wire path) - every case of this file runs both ways."""
    monkeypatch.setenv("GPX_WD_TILE", "256" if request.param.startswith("256") else "512")


@pytest.mark.parametrize("name", [n for n in dir(scen) if n.startswith("test_") and "oracle_lib" in
                                  getattr(scen, n).__code__.co_varnames[:getattr(scen, n).__code__.co_argcount]])
def test_oracle_scenarios_on_the_engine(hip_lib, name):
    """Every known-answer scenario that pins the oracle, run on the HIP library instead."""
    getattr(scen, name)(hip_lib)


@pytest.mark.parametrize("seed,damage", [(1, 0.0), (2, 0.3), (3, 0.6)])
def test_decode_fuzz(hip_lib, oracle_lib, seed, damage):
    rng = np.random.default_rng(seed)
    ((eh, wh), (eo, wo)), names = make_wire_pair(hip_lib, oracle_lib, 1500, 3, rng)
    for burst in range(4):
        frames = random_frames(names, 3000, rng, damage)
        assert_same_decode(wh.decode(frames), wo.decode(frames), f"burst {burst}")
    # tight capacities: overflow is reported frame by frame, never written
    frames = random_frames(names, 500, rng, 0.0)
    assert_same_decode(wh.decode(frames, 40, 30, 20, 10), wo.decode(frames, 40, 30, 20, 10), "tight")


def test_decode_many_tiles(hip_lib, oracle_lib):
    """A burst of 420,000 frames of every class (damaged ones included) = 1,641 tiles of 256 frames,
    more than the chip holds at once: the one-launch decode's offsets come from a look-back over
    tiles that are still running (gpx_wire.hip.h, k_wire_decode1) - same records, same order, same
    counts as the oracle, call after call (the look-back words carry the call's epoch)."""
    rng = np.random.default_rng(77)
    ((eh, wh), (eo, wo)), names = make_wire_pair(hip_lib, oracle_lib, 1500, 3, rng, max_batch=1 << 19)
    for rep, damage in ((140, 0.2), (97, 0.0)):
        buf, off = W.concat_frames(random_frames(names, 3000, rng, damage))
        big = np.tile(buf, rep)
        boff = (off[None, :-1] + (np.arange(rep, dtype=np.int64) * off[-1])[:, None]).reshape(-1)
        boff = np.concatenate([boff, np.array([off[-1] * rep], np.int64)])
        a, b = wh.decode(None, buf_off=(big, boff)), wo.decode(None, buf_off=(big, boff))
        assert_same_decode(a, b, f"x{rep}")
        assert a.counts["n_votes"] + a.counts["n_commits"] + a.counts["n_accepts"] + a.counts["n_requests"] > 3000 * rep // 4


def test_names_churn_and_table_rebuild(hip_lib, oracle_lib):
    """bind / unbind cycles far beyond the tombstone threshold: lookups stay exact."""
    G = 512
    rng = np.random.default_rng(9)
    eh, eo = Engine(hip_lib, 100, G, kmax=3, window=8), Engine(oracle_lib, 100, G, kmax=3, window=8)
    wh, wo = W.WireEngine(eh), W.WireEngine(eo)
    live = {}
    for it in range(30):
        rows = rng.choice(G, size=200, replace=False).astype(np.int32)
        names = [b"n%d_%d" % (it, int(r)) for r in rows]
        sa, sb = wh.bind(names, rows), wo.bind(names, rows)
        assert sa.tolist() == sb.tolist()
        for r, nm, s in zip(rows, names, sa):
            if s == S_OK:
                live[int(r)] = nm
        drop = rng.choice(G, size=150, replace=False).astype(np.int32)
        assert wh.unbind(drop).tolist() == wo.unbind(drop).tolist()
        for r in drop:
            live.pop(int(r), None)
        probe = [live[r] for r in sorted(live)] + [b"absent%d" % it]
        la, lb = wh.lookup(probe), wo.lookup(probe)
        assert la.tolist() == lb.tolist() == sorted(live) + [-1]


def test_pack_commits_matches_oracle_and_roundtrips(hip_lib, oracle_lib):
    """Decisions of real accept-reply rounds (adversarial mix: preempts, several slots per group)
    packed into BATCHED_COMMIT frames: bytes identical to the oracle's; decoding them on an
    acceptor replica yields exactly the decided (group, slot) set."""
    G, k = 4096, 3
    rng = np.random.default_rng(4)
    members = [100, 101, 102]
    (eh, wh), (eo, wo) = [(e, W.WireEngine(e)) for e in
                          (Engine(hip_lib, 100, G, kmax=k, window=8, max_batch=1 << 16),
                           Engine(oracle_lib, 100, G, kmax=k, window=8))]
    names = [b"grp-%d" % g for g in range(G)]
    mem = np.tile(np.array(members, np.int32), (G, 1))
    acc = Engine(hip_lib, 101, G, kmax=k, window=8, max_batch=1 << 16)  # a remote acceptor replica
    wacc = W.WireEngine(acc)
    for e, we in ((eh, wh), (eo, wo), (acc, wacc)):
        assert (e.create_groups(np.arange(G), mem, k, hri_create(G, k, 100)) == S_OK).all()
        assert (we.bind(names, np.arange(G)) == S_OK).all()
    for r in range(3):
        g = np.arange(G, dtype=np.int32)
        for _ in range(2):  # two outstanding slots per group and round
            eh.propose(g), eo.propose(g)
        cols = [np.concatenate(c) for c in zip(
            streams.vote_round(G, members, 2 * r, 100, mix=True),
            streams.vote_round(G, members, 2 * r + 1, 100, mix=True))]
        order = rng.permutation(cols[0].shape[0])
        cols = [np.ascontiguousarray(c[order]) for c in cols]
        dh, do = eh.accept_reply(*cols), eo.accept_reply(*cols)
        assert (dh.as_tuple_array() == do.as_tuple_array()).all()
        fh, gh, nbh = wh.pack_commits(dh)
        fo, go, nbo = wo.pack_commits(do)
        assert gh.tolist() == go.tolist() and nbh == nbo
        assert fh == fo
        dec = wacc.decode(fh)
        assert (dec.f_status == W.W_OK).all() and dec.f_gidx.tolist() == gh.tolist()
        want = sorted(zip(dh.gidx[dh.kind == D_DECISION].tolist(), dh.slot[dh.kind == D_DECISION].tolist()))
        got = sorted(zip(dec.commits["gidx"].tolist(), dec.commits["slot"].tolist()))
        assert got == want


@pytest.mark.parametrize("lead", [1, 2, 3, 4, 7, 12, 13])
def test_decode_burst_at_any_byte_offset(hip_lib, oracle_lib, lead):
    """The burst need not start on a 16-byte (or even 4-byte) boundary: the staging copy aligns its
    16-byte loads on the address and shifts the LDS image by the same lead (wire_stage), and never
    reads before the first frame's dword."""
    rng = np.random.default_rng(40 + lead)
    ((eh, wh), (eo, wo)), names = make_wire_pair(hip_lib, oracle_lib, 600, 3, rng)
    buf, off = W.concat_frames(random_frames(names, 1500, rng, 0.1))
    shifted = np.concatenate([rng.integers(0, 256, lead).astype(np.uint8), buf])
    assert_same_decode(wh.decode(None, buf_off=(shifted, off + lead)), wo.decode(None, buf_off=(shifted, off + lead)),
                       f"lead {lead}")


def test_pack_commits_many_tiles(hip_lib, oracle_lib):
    """300,000 groups' decisions (two slots for every third group, preemptions that open no frame) =
    1,5xx tiles of 256 rows: the one-launch encoder's byte and frame offsets come from a look-back
    over tiles still running (k_pack_commits1) - bytes, frame order and totals identical to the
    oracle's; a block of more than 256 consecutive rows of one group is refused by both."""
    from gigapaxos_amd import GpxError
    from gigapaxos_amd._abi import Decisions
    G, k = 300_000, 3
    rng = np.random.default_rng(12)
    mem = np.tile(np.array([100, 101, 102], np.int32), (G, 1))
    names = [bytes(r) for r in W.fixed_names(np.arange(G))]
    pair = []
    for lib in (hip_lib, oracle_lib):
        e = Engine(lib, 100, G, kmax=k, window=8, max_batch=1 << 19)
        we = W.WireEngine(e)
        assert (e.create_groups(np.arange(G), mem, k, hri_create(G, k, 100)) == S_OK).all()
        assert (we.bind(names, np.arange(G)) == S_OK).all()
        pair.append(we)
    reps = 1 + (np.arange(G) % 3 == 0)
    gidx = np.repeat(np.arange(G, dtype=np.int32), reps)
    n = gidx.shape[0]
    slot = (7 + np.arange(n) % 2).astype(np.int32)
    kind = np.where(rng.random(n) < 0.03, 1, D_DECISION).astype(np.uint8)  # 1 = PREEMPTED: no frame
    dec = Decisions(gidx, slot, np.zeros(n, np.int32), np.full(n, 100, np.int32),
                    rng.integers(0, 5, n).astype(np.int32), kind, np.zeros(0, np.uint8))
    (fh, gh, nbh), (fo, go, nbo) = [we.pack_commits(dec) for we in pair]
    assert nbh == nbo and gh.tolist() == go.tolist() and len(fh) > G * 9 // 10
    assert fh == fo
    hot = Decisions(np.full(300, 5, np.int32), np.arange(300, dtype=np.int32), np.zeros(300, np.int32),
                    np.full(300, 100, np.int32), np.zeros(300, np.int32), np.full(300, D_DECISION, np.uint8),
                    np.zeros(0, np.uint8))
    for we in pair:
        with pytest.raises(GpxError):
            we.pack_commits(hot)


@pytest.mark.parametrize("seed", [1, 2])
def test_pack_accept_replies_fuzz(hip_lib, oracle_lib, seed):
    """Random accept batches (several slots and ballots per group, NACKs, dropped accepts, a hot
    group beyond the per-call limits, unnamed rows): BATCHED_ACCEPT_REPLY bytes, destinations and
    the unbatched flags identical to the oracle's."""
    rng = np.random.default_rng(seed)
    ((eh, wh), (eo, wo)), names = make_wire_pair(hip_lib, oracle_lib, 1500, 5, rng)
    G = 1500
    for it in range(4):
        n = 6000
        g = rng.integers(-1, G + 1, n).astype(np.int32)
        g[rng.random(n) < 0.25] = 77        # hot group: > 256 replies in one call
        g[rng.random(n) < 0.05] = 78        # > 16 records: cooperative sort path
        slot = rng.integers(1, 40, n).astype(np.int32)
        bnum = rng.choice([0, 0, 0, 1, 2, 3, 4, 5], size=n).astype(np.int32)
        bcoord = rng.integers(100, 105, n).astype(np.int32)
        sender = np.where(rng.random(n) < 0.9, bcoord, 100).astype(np.int32)
        maxcp = rng.integers(-1, 30, n).astype(np.int32)
        status = rng.choice([0, 0, 0, 0, 2, 3], size=n).astype(np.uint8)
        status[(g < 0) | (g >= G)] = 1
        req = rng.integers(-2**62, 2**62, n)
        use_opt = it % 2 == 0
        a = wh.pack_accept_replies(g, slot, bnum, bcoord, maxcp, status, sender if use_opt else None,
                                   req if use_opt else None)
        b = wo.pack_accept_replies(g, slot, bnum, bcoord, maxcp, status, sender if use_opt else None,
                                   req if use_opt else None)
        assert a[1].tolist() == b[1].tolist() and a[2].tolist() == b[2].tolist(), f"iter {it} frame table"
        assert a[3].tolist() == b[3].tolist(), f"iter {it} unbatched"
        assert a[4] == b[4] and a[0] == b[0], f"iter {it} bytes"
        assert eh.counters() == eo.counters()


def test_frame_level_cluster_matches_oracle(hip_lib, oracle_lib):
    """Three replicas exchanging nothing but wire frames (tests/wire_cluster.py), engine cluster vs
    oracle cluster: every frame on the wire and every execution log identical."""
    from tests.wire_cluster import WireCluster
    G, R = 3000, 6
    names = [b"service/%d" % g for g in range(G)]
    rng = np.random.default_rng(8)
    coord = rng.choice([100, 101, 102], size=G).astype(np.int32)
    ch = WireCluster(hip_lib, [100, 101, 102], names, coord)
    co = WireCluster(oracle_lib, [100, 101, 102], names, coord)
    for r in range(R):
        groups = rng.permutation(G)[: G - 100 * r]
        dh, do = ch.round(groups, r), co.round(groups, r)
        for c in dh:
            assert (dh[c].as_tuple_array() == do[c].as_tuple_array()).all()
    assert ch.trace == co.trace
    for nid in (100, 101, 102):
        assert ch.executed(nid).tolist() == co.executed(nid).tolist()
        sh, so = ch.eng[nid].snapshot(np.arange(G))[0], co.eng[nid].snapshot(np.arange(G))[0]
        assert sh.tobytes() == so.tobytes()


def test_wire_codec_against_java_reading(hip_lib):
    """tests/test_wire_model.py (the Python reading of BatchedAcceptReply.java:103-173, BatchedCommit.java:156-215,
    PaxosPacketDemultiplexerFast.java:66-103) on the engine, on each decode tiling (the autouse fixture)"""
    from tests import test_wire_model as T
    T.test_decode_of_damaged_bursts_against_java_reading(hip_lib, 2000, 30_000, 12, 0.5)
    T.test_decode_of_damaged_bursts_against_java_reading(hip_lib, 50, 20_000, 13, 0.9)
    T.test_pack_of_random_batches_against_java_reading(hip_lib, 3000, 22)
