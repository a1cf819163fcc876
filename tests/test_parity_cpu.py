"""CPU-side checks of the differential harness itself (oracle vs a second oracle instance) and of
the host logic that does not need a GPU.  The real parity tests are in test_parity_gpu.py."""
import numpy as np

from gigapaxos_amd import streams
from tests.parity_common import make_pair, create_mixed_groups, fuzz, wrap32


def test_harness_selfcheck(oracle_lib):
    rng = np.random.default_rng(11)
    ea, eb = make_pair(oracle_lib, oracle_lib, 100, 48, 5, 64)
    create_mixed_groups(ea, eb, 48, 5, [100, 101, 102, 103, 104, 105], rng)
    fuzz(ea, eb, 48, [100, 101, 102, 103, 104, 105], rng, steps=300, batch=200)
    c = ea.counters()
    assert c[0] > 0 and c[1] > 0 and c[2] > 0  # votes, outputs and drops all exercised


def test_harness_wraparound(oracle_lib):
    """slots straddling Integer.MAX_VALUE -> MIN_VALUE (the `a - b < 0` idiom everywhere)."""
    rng = np.random.default_rng(12)
    base = (1 << 31) - 15
    ea, eb = make_pair(oracle_lib, oracle_lib, 100, 16, 3, 64)
    create_mixed_groups(ea, eb, 16, 3, [100, 101, 102, 103], rng, slot_base=base)
    fuzz(ea, eb, 16, [100, 101, 102, 103], rng, steps=200, batch=100, slot_base=base)
    assert int(wrap32(base + 20)) < 0


def test_vote_round_shape_and_determinism():
    cols = streams.vote_round(1000, [100, 101, 102], 4, 100)
    assert all(c.dtype == np.int32 and c.shape == (3000,) for c in cols)
    cols2 = streams.vote_round(1000, [100, 101, 102], 4, 100)
    assert all((a == b).all() for a, b in zip(cols, cols2))
    g, bnum, bcoord, slot, acc, mcp = cols
    assert (np.bincount(g, minlength=1000) == 3).all()
    assert (slot == 5).all() and (mcp == 4).all() and (bcoord == 100).all()
    # every group hears from each member exactly once
    key = g.astype(np.int64) * 1000 + acc
    assert np.unique(key).shape[0] == 3000
    mix = streams.vote_round(1000, [100, 101, 102], 4, 100, mix=True)
    assert mix[0].shape[0] == 3000 + 30 + 15 + 3


def test_clean_round_decides_every_group(oracle_lib):
    from gigapaxos_amd import Engine, hri_create, D_DECISION
    G = 2000
    e = Engine(oracle_lib, 100, G, kmax=3, window=8)
    e.create_groups(np.arange(G), np.tile(np.array([100, 101, 102], np.int32), (G, 1)), 3,
                    hri_create(G, 3, 100))
    for r in range(5):
        e.propose(np.arange(G))
        d = e.accept_reply(*streams.vote_round(G, [100, 101, 102], r, 100))
        assert d.gidx.shape[0] == G and (d.kind == D_DECISION).all()
        assert (np.sort(d.gidx) == np.arange(G)).all()
        assert (d.slot == r + 1).all()
        # createHRI rows: nodeSlots start at 0 and two of three voters reported r
        assert (d.median_cp == r).all()


def test_shard_hash():
    g = np.arange(1 << 16)
    s = streams.shard_of(g, 8)
    cnt = np.bincount(s, minlength=8)
    assert cnt.min() > 7500 and cnt.max() < 8900
    assert int(streams.fmix32(np.array([1], np.uint32))[0]) == 0x514E28B7  # murmur3 fmix32(1)


def test_config5_churn_properties_on_oracle(oracle_lib):
    """BASELINE config #5 driver on the oracle alone (the GPU test compares the engine with it):
    every live group decides every round, late votes / proposals for retired groups are dropped
    with NOGROUP, retired rows come back as HotRestoreInfo and are reusable."""
    from tests.parity_common import churn_run
    from gigapaxos_amd import Engine, S_NOGROUP, HRI_DTYPE
    G_live, cap, R, k = 2000, 2000 + 3 * 20, 6, 5
    e = Engine(oracle_lib, 100, cap, kmax=k, window=8)
    (out,), live = churn_run([e], G_live, cap, R, k, seed=7)
    for r, (dec, vst, pout, pst, rows, st_ret, st_new) in enumerate(out):
        assert dec.shape[0] == G_live and (st_ret == 0).all() and (st_new == 0).all()
        assert (vst == S_NOGROUP).sum() == (k * 20 if r else 0)
        assert (pst == S_NOGROUP).sum() == (20 if r else 0)
        hri = np.frombuffer(rows, dtype=HRI_DTYPE)
        assert (hri["has_coord"] == 1).all() and (hri["acc_slot"] >= 1).all()
    assert len(set(live.tolist())) == G_live
