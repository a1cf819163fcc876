"""N > 1 path on CPU: world size 2 over gloo.  Each rank owns one hash shard of a global group
space (its own engine — here the oracle, since there is no GPU), routes its part of every batch,
and the ranks exchange only telemetry (decision counts / load counters) — no collective carries
protocol data.  The union of the two shards' decisions must equal the single-engine result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gigapaxos_amd import Engine, hri_create, streams, S_OK
from gigapaxos_amd.sharding import ShardMap

G, K, ROUNDS = 3000, 3, 4
MEMBERS = [100, 101, 102]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _single_engine_reference(lib):
    e = Engine(lib, 100, G, kmax=K, window=8)
    e.create_groups(np.arange(G), np.tile(np.array(MEMBERS, np.int32), (G, 1)), K, hri_create(G, K, 100))
    out = []
    for r in range(ROUNDS):
        e.propose(np.arange(G, dtype=np.int32))
        d = e.accept_reply(*streams.vote_round(G, MEMBERS, r, 100, mix=True))
        out.append(d.as_tuple_array())
    return out


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.oracle_binding import load_oracle

    lib = load_oracle()
    sm = ShardMap(G, world)
    gl = sm.globals_of[rank]
    nloc = gl.shape[0]
    e = Engine(lib, 100, nloc, kmax=K, window=8)
    st = e.create_groups(np.arange(nloc), np.tile(np.array(MEMBERS, np.int32), (nloc, 1)), K,
                         hri_create(nloc, K, 100))
    assert (st == S_OK).all()
    per_round = []
    total = 0
    for r in range(ROUNDS):
        e.propose(np.arange(nloc, dtype=np.int32))
        cols, _ = sm.route(streams.vote_round(G, MEMBERS, r, 100, mix=True), rank)
        d = e.accept_reply(*cols)
        arr = d.as_tuple_array()
        arr[:, 0] = sm.to_global(rank, arr[:, 0])
        per_round.append(arr)
        total += arr.shape[0]
    # telemetry only: decision totals and the load counters, max-over-ranks of a (fake) time
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    cnt = torch.tensor([total], dtype=torch.int64)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    ctr = torch.tensor(e.counters(), dtype=torch.int64)
    allc = [torch.zeros_like(ctr) for _ in range(world)]
    dist.all_gather(allc, ctr)
    dist.barrier()
    q.put((rank, per_round, int(cnt.item()), float(t.item()), [c.tolist() for c in allc]))
    dist.destroy_process_group()


def test_two_shards_equal_single_engine(oracle_lib):
    ref = _single_engine_reference(oracle_lib)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, per_round, total, tmax, allc = q.get(timeout=120)
        res[rank] = (per_round, total, tmax, allc)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == sum(a.shape[0] for a in ref)
    assert res[0][2] == res[1][2] == 2.0
    assert res[0][3] == res[1][3]
    assert sum(c[0] for c in res[0][3]) == sum(streams.vote_round(G, MEMBERS, r, 100, mix=True)[0].shape[0]
                                               for r in range(ROUNDS))
    for r in range(ROUNDS):
        merged = np.concatenate([res[0][0][r], res[1][0][r]])
        # same multiset of decisions; within a shard the arrival order is preserved
        key = lambda a: a[np.lexsort(a.T[::-1])]  # noqa: E731
        assert (key(merged) == key(ref[r])).all()
        for rk in (0, 1):
            sm = ShardMap(G, 2)
            mine = ref[r][sm.shard[ref[r][:, 0]] == rk]
            assert (mine == res[rk][0][r]).all()


def test_shard_map_roundtrip():
    sm = ShardMap(10000, 8)
    assert sm.counts.sum() == 10000 and sm.counts.min() > 1000
    for s in range(8):
        loc = np.arange(sm.counts[s])
        assert (sm.shard[sm.to_global(s, loc)] == s).all()
        assert (sm.local[sm.to_global(s, loc)] == loc).all()
    cols = [np.array([5, -1, 9999, 10000, 7], np.int32), np.arange(5, dtype=np.int32)]
    seen = 0
    for s in range(8):
        out, idx = sm.route(cols, s)
        seen += out[0].shape[0]
        for gl, orig in zip(out[0], idx):
            if cols[0][orig] < 0 or cols[0][orig] >= 10000:
                assert gl == -1 and s == 0
            else:
                assert sm.to_global(s, [gl])[0] == cols[0][orig]
    assert seen == 5
