"""gpx_route_batch_dev (device = fmix32(gidx) % n_shards, binning on the device) against
the host-side ShardMap.route: same records per shard, in the batch's order, gidx rewritten to the
shard-local dense index, out-of-range indices on shard 0 as -1."""
import numpy as np
import pytest

from gigapaxos_amd import Engine
from gigapaxos_amd.sharding import ShardMap

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_shards,n,G", [(1, 5000, 3000), (2, 40000, 3000), (8, 300000, 100000), (16, 70001, 777)])
def test_route_matches_shard_map(hip_lib, n_shards, n, G):
    import torch

    rng = np.random.default_rng(n_shards * 1000 + n)
    dev = torch.device("cuda:0")
    e = Engine(hip_lib, 100, 64, kmax=3, window=8, max_batch=n + 16)
    sm = ShardMap(G, n_shards)
    g = rng.integers(0, G, n).astype(np.int32)
    bad = rng.random(n) < 0.01
    g[bad] = rng.choice([-1, G, G + 9, -(1 << 31)], size=int(bad.sum()))
    cols = [g] + [rng.integers(-(1 << 31), (1 << 31) - 1, n).astype(np.int32) for _ in range(5)]
    d_in = [torch.from_numpy(c).to(dev) for c in cols]
    d_out = [torch.empty(n, dtype=torch.int32, device=dev) for _ in cols]
    g2l = torch.from_numpy(sm.local).to(dev)
    off = torch.zeros(n_shards + 1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    e.route_dev(n, [t.data_ptr() for t in d_in], g2l.data_ptr(), G, n_shards, [t.data_ptr() for t in d_out],
                off.data_ptr())
    e.sync()
    off_h = off.cpu().numpy()
    assert off_h[0] == 0 and off_h[-1] == n
    out_h = [t.cpu().numpy() for t in d_out]
    for s in range(n_shards):
        want, _ = sm.route(cols, s)
        lo, hi = int(off_h[s]), int(off_h[s + 1])
        assert hi - lo == want[0].shape[0], f"shard {s}: count"
        for k in range(len(cols)):
            assert (out_h[k][lo:hi] == want[k]).all(), f"shard {s} column {k}"
    e.close()
