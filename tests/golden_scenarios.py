"""Seeded scenarios whose outputs are frozen under tests/golden/ (see tests/golden/make_golden.py).
Each takes a library behind the ABI and returns a dict of arrays."""
import numpy as np

from gigapaxos_amd import Engine, hri_create, streams, S_OK
from gigapaxos_amd import wire as W
from gigapaxos_amd.loopback import LoopbackCluster
from tests.wire_common import make_wire_pair, random_frames


def decided_stream(lib):
    """Config #3's adversarial vote stream on 4096 groups, 5 rounds: the decided stream, vote
    statuses and the final HotRestoreInfo rows."""
    G, members = 4096, [100, 101, 102]
    e = Engine(lib, 100, G, kmax=3, window=8, max_batch=1 << 15)
    mem = np.tile(np.array(members, np.int32), (G, 1))
    assert (e.create_groups(np.arange(G), mem, 3, hri_create(G, 3, 100)) == S_OK).all()
    dec, st = [], []
    for r in range(5):
        e.propose(np.arange(G, dtype=np.int32))
        d = e.accept_reply(*streams.vote_round(G, members, r, 100, mix=True))
        dec.append(d.as_tuple_array())
        st.append(d.status)
    rows = e.snapshot(np.arange(G))[0]
    e.close()
    return {"decisions": np.concatenate(dec), "n_per_round": np.array([x.shape[0] for x in dec]),
            "status": np.concatenate(st), "rows": np.frombuffer(rows.tobytes(), np.uint8)}


def full_pipeline(lib):
    """Config #2's shape (3 replicas, coordinators spread, propose -> accept -> reply -> decide ->
    commit -> execute) on 600 groups, 6 rounds: execution logs of every replica."""
    G = 600
    rng = np.random.default_rng(7)
    coord = rng.choice([100, 101, 102], size=G).astype(np.int32)
    c = LoopbackCluster(lib, [100, 101, 102], G, window=8, max_batch=1 << 14, coordinator=coord)
    decs = []
    for r in range(6):
        decs.append(c.round(rng.permutation(G).astype(np.int32)))
    out = {"decisions": np.concatenate(decs)}
    for nid in (100, 101, 102):
        out["exec_%d" % nid] = c.executed(nid)
    c.close()
    return out


def wire_decode(lib):
    """A burst of 2000 frames, 40 % of them damaged: every decode output."""
    rng = np.random.default_rng(5)
    ((e, we), (e2, we2)), names = make_wire_pair(lib, lib, 700, 3, rng)
    d = we.decode(random_frames(names, 2000, rng, 0.4))
    out = {"f_status": d.f_status, "f_gidx": d.f_gidx, "f_type": d.f_type,
           "counts": np.array([d.counts[k] for k in sorted(d.counts)])}
    for cls in ("votes", "commits", "accepts", "requests"):
        for k, v in getattr(d, cls).items():
            out[cls + "_" + k] = v
    e.close()
    e2.close()
    return out


SCENARIOS = {"decided_stream": decided_stream, "full_pipeline": full_pipeline, "wire_decode": wire_decode}
