"""Edges of the batch calls on the HIP engine: empty batches, capacity errors, tile-boundary batch
sizes, unaligned device columns (scalar-load path), two engines side by side."""
import numpy as np
import pytest

from gigapaxos_amd import Engine, GpxError, hri_create, streams, S_OK
from gigapaxos_amd import wire as W
from tests.parity_common import make_pair

pytestmark = pytest.mark.gpu


def _mk(lib, G=3000, k=3, my_id=100, max_batch=1 << 15, window=8):
    e = Engine(lib, my_id, G, kmax=k, window=window, max_batch=max_batch)
    mem = np.tile(np.arange(100, 100 + k, dtype=np.int32), (G, 1))
    assert (e.create_groups(np.arange(G), mem, k, hri_create(G, k, 100)) == S_OK).all()
    return e


def test_empty_batches_and_capacity(hip_lib):
    e = _mk(hip_lib, G=64, max_batch=128)
    we = W.WireEngine(e)
    z = np.zeros(0, np.int32)
    assert all(x.shape[0] == 0 for x in e.propose(z))
    assert e.accept_reply(z, z, z, z, z, z).gidx.shape[0] == 0
    assert e.accept(z, z, z, z, z)[1].gidx.shape[0] == 0
    assert e.commit(z, z, z, z, z)[1].gidx.shape[0] == 0
    assert e.prepare(z, z, z, z)[1] == []
    assert we.decode([]).counts["n_votes"] == 0
    assert W.request_batch(we, z, z)[2]["gidx"].shape[0] == 0
    big = np.zeros(129, np.int32)
    for call in (lambda: e.propose(big), lambda: e.accept_reply(big, big, big, big, big, big),
                 lambda: e.commit(big, big, big, big, big), lambda: e.prepare(big, big, big, big),
                 lambda: W.request_batch(we, big, big)):
        with pytest.raises(GpxError):
            call()
    assert e.counters() == (0, 0, 0)
    e.close()


@pytest.mark.parametrize("n", [1, 63, 64, 4095, 4096, 4097, 8192, 12287, 12288, 12289, 20000])
def test_tile_boundary_batch_sizes(hip_lib, oracle_lib, n):
    """Batch sizes around the 4096-record scatter tiles and the multi-tile histogram workgroups."""
    G = 3000
    eh, eo = make_pair(hip_lib, oracle_lib, 100, G, 3, 8, max_batch=1 << 15)
    mem = np.tile(np.array([100, 101, 102], np.int32), (G, 1))
    for e in (eh, eo):
        assert (e.create_groups(np.arange(G), mem, 3, hri_create(G, 3, 100)) == S_OK).all()
        e.propose(np.arange(G, dtype=np.int32))
    cols = streams.vote_round(G, [100, 101, 102], 0, 100, mix=True)
    rng = np.random.default_rng(n)
    pick = rng.integers(0, cols[0].shape[0], n)
    cols = [np.ascontiguousarray(c[pick]) for c in cols]
    dh, do = eh.accept_reply(*cols), eo.accept_reply(*cols)
    assert (dh.as_tuple_array() == do.as_tuple_array()).all() and (dh.status == do.status).all()
    assert eh.snapshot(np.arange(G))[0].tobytes() == eo.snapshot(np.arange(G))[0].tobytes()


def test_unaligned_device_columns(hip_lib, oracle_lib):
    """Device columns that are not 16-byte aligned take the scalar-load kernels: same answer."""
    import torch

    G, n = 3000, 9001
    eh, eo = make_pair(hip_lib, oracle_lib, 100, G, 3, 8, max_batch=1 << 15)
    mem = np.tile(np.array([100, 101, 102], np.int32), (G, 1))
    for e in (eh, eo):
        assert (e.create_groups(np.arange(G), mem, 3, hri_create(G, 3, 100)) == S_OK).all()
        e.propose(np.arange(G, dtype=np.int32))
    cols = streams.vote_round(G, [100, 101, 102], 0, 100, mix=True)
    cols = [np.ascontiguousarray(c[:n]) for c in cols]
    do = eo.accept_reply(*cols)
    dev = torch.device("cuda:0")
    pads = [torch.zeros(n + 8, dtype=torch.int32, device=dev) for _ in range(6)]
    views = []
    for p, c in zip(pads, cols):
        v = p[1:n + 1]  # 4-byte offset: not 16-byte aligned
        v.copy_(torch.from_numpy(c))
        assert v.data_ptr() % 16 != 0
        views.append(v)
    outs = [torch.zeros(n + 1, dtype=torch.int32, device=dev)[1:] for _ in range(5)]
    kind = torch.zeros(n + 1, dtype=torch.uint8, device=dev)[1:]
    st = torch.zeros(n + 1, dtype=torch.uint8, device=dev)[1:]
    n_out = torch.zeros(1, dtype=torch.int32, device=dev)
    eh.call_dev("accept_reply_batch", n, *[v.data_ptr() for v in views], *[o.data_ptr() for o in outs],
                kind.data_ptr(), n_out.data_ptr(), st.data_ptr())
    eh.sync()
    m = int(n_out)
    got = np.stack([o[:m].cpu().numpy() for o in outs] + [kind[:m].cpu().numpy().astype(np.int32)], axis=1)
    assert m == do.gidx.shape[0] and (got == do.as_tuple_array()).all()
    assert (st.cpu().numpy() == do.status).all()


def test_two_engines_side_by_side(hip_lib, oracle_lib):
    """Two engine handles on one GPU (two replicas of a node-local test cluster, or two shards):
    interleaved calls do not disturb each other."""
    G = 2000
    ea, eb = _mk(hip_lib, G, my_id=100), _mk(hip_lib, G, my_id=101)
    oa, ob = _mk(oracle_lib, G, my_id=100), _mk(oracle_lib, G, my_id=101)
    g = np.arange(G, dtype=np.int32)
    for r in range(3):
        for e in (ea, oa):
            e.propose(g)
        cols = streams.vote_round(G, [100, 101, 102], r, 100)
        acc = [np.ascontiguousarray(c) for c in (g, np.zeros(G, np.int32), np.full(G, 100, np.int32),
                                                 np.full(G, r + 1, np.int32), np.zeros(G, np.int32))]
        (ra, xa), (rb, xb) = eb.accept(*acc), ob.accept(*acc)
        da, db = ea.accept_reply(*cols), oa.accept_reply(*cols)
        assert all((x == y).all() for x, y in zip(ra, rb))
        assert (da.as_tuple_array() == db.as_tuple_array()).all()
    for e, o in ((ea, oa), (eb, ob)):
        assert e.snapshot(g)[0].tobytes() == o.snapshot(g)[0].tobytes()
