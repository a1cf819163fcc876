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
