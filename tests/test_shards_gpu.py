"""BASELINE config #4 on the HIP engine: ONE space of 1,000,000 groups x 5 replicas, hash-sharded
(`fmix32(gidx) % n`) over N independent engines - here all N on cuda:0, each on its own
stream, because the builder's box has one GPU; on an 8-GPU node each engine is one rank's.  Every
round's batch is binned ON THE DEVICE (`gpx_route_batch_dev`), each engine sees only its shard's
records with shard-local group indices, and nothing is exchanged between the engines (groups are
independent: PaxosManager.java:3170-3171).  The union of the shards' decided streams, per-vote statuses,
proposal answers and HotRestoreInfo rows must equal a single HIP engine's over the whole space, which
must equal the oracle's.  All engine calls go through the `_dev` entry points bench.py times."""
import numpy as np
import pytest

from gigapaxos_amd import Engine, ORDERED_PROPOSE, hri_create, streams, S_NOGROUP, S_OK
from gigapaxos_amd.sharding import ShardMap

pytestmark = pytest.mark.gpu

G, K, R = 1_000_000, 5, 3
MEMBERS = list(range(100, 100 + K))


class DevEngine:
    """One HIP engine driven through the device-pointer calls on its own torch stream."""

    def __init__(self, lib, groups, max_batch, torch):
        self.torch, self.G = torch, groups
        self.dev = torch.device("cuda:0")
        self.e = Engine(lib, 100, groups, kmax=K, window=8, max_batch=max_batch)
        self.ts = torch.cuda.Stream(device=self.dev)
        self.e.set_stream(self.ts.cuda_stream)
        self.e.set_ordered_batches(ORDERED_PROPOSE)
        mem = np.tile(np.array(MEMBERS, np.int32), (groups, 1))
        assert (self.e.create_groups(np.arange(groups, dtype=np.int32), mem, K, hri_create(groups, K, 100)) == S_OK).all()
        i32 = lambda n: torch.empty(n, dtype=torch.int32, device=self.dev)  # noqa: E731
        u8 = lambda n: torch.empty(n, dtype=torch.uint8, device=self.dev)  # noqa: E731
        self.g_all = torch.arange(groups, dtype=torch.int32, device=self.dev)
        self.p = [i32(groups) for _ in range(4)] + [u8(groups)]
        self.d = [i32(max_batch) for _ in range(5)] + [u8(max_batch)]
        self.n_out = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self.st = u8(max_batch)

    def propose_all(self):
        self.e.call_dev("propose_batch", self.G, self.g_all.data_ptr(), 0, *[t.data_ptr() for t in self.p])

    def accept_reply(self, n, col_ptrs):
        self.nv = n
        self.e.call_dev("accept_reply_batch", n, *col_ptrs, *[t.data_ptr() for t in self.d], self.n_out.data_ptr(),
                        self.st.data_ptr())

    def results(self):
        """(propose columns, decisions as rows, per-vote status) of the round just issued."""
        self.e.sync()
        m = int(self.n_out.item())
        dec = np.stack([t[:m].cpu().numpy().astype(np.int32) for t in self.d], axis=1)
        return [t.cpu().numpy() for t in self.p], dec, self.st[: self.nv].cpu().numpy()

    def close(self):
        self.e.sync()
        self.e.close()


@pytest.fixture(scope="module")
def whole_space(hip_lib, oracle_lib):
    """The rounds of config #4's stream (adversarial mix, a few indices outside the table), what the oracle and ONE
    HIP engine over the whole space answer: both must agree before any shard is looked at."""
    import torch

    rng = np.random.default_rng(44)
    rounds = []
    for r in range(R):
        cols = [c.copy() for c in streams.vote_round(G, MEMBERS, r, 100, config_id=4, mix=True)]
        bad = rng.integers(0, cols[0].shape[0], 64)
        cols[0][bad] = rng.choice([-1, G, G + 77, -(1 << 31)], size=64)
        rounds.append(cols)
    nv_max = max(c[0].shape[0] for c in rounds)
    eo = Engine(oracle_lib, 100, G, kmax=K, window=8)
    mem = np.tile(np.array(MEMBERS, np.int32), (G, 1))
    assert (eo.create_groups(np.arange(G, dtype=np.int32), mem, K, hri_create(G, K, 100)) == S_OK).all()
    one = DevEngine(hip_lib, G, nv_max + 4096, torch)
    g = np.arange(G, dtype=np.int32)
    ref = []
    for r, cols in enumerate(rounds):
        po = eo.propose(g)
        do = eo.accept_reply(*cols)
        dcols = [torch.from_numpy(c).to(one.dev) for c in cols]
        torch.cuda.synchronize()
        one.propose_all()
        one.accept_reply(cols[0].shape[0], [t.data_ptr() for t in dcols])
        ph, dh, sh = one.results()
        for x, y, nm in zip(ph, po, ("slot", "bnum", "bcoord", "median", "status")):
            assert (x == y).all(), f"round {r}: single HIP engine vs oracle, propose {nm}"
        assert dh.shape == do.as_tuple_array().shape and (dh == do.as_tuple_array()).all(), f"round {r}: decisions"
        assert (sh == do.status).all(), f"round {r}: per-vote status"
        assert (sh == S_NOGROUP).sum() >= 60
        ref.append((ph, dh, sh))
    rows = one.e.snapshot(g)[0]
    assert rows.tobytes() == eo.snapshot(g)[0].tobytes()
    counters = one.e.counters()
    assert counters == eo.counters()
    one.close()
    eo.close()
    return rounds, ref, rows, counters


@pytest.mark.parametrize("n_shards", [2, 8])
def test_config4_hash_sharded_hip_engines_equal_the_single_engine(hip_lib, whole_space, n_shards):
    import torch

    rounds, ref, rows_one, counters_one = whole_space
    dev = torch.device("cuda:0")
    sm = ShardMap(G, n_shards)
    nv_max = max(c[0].shape[0] for c in rounds)
    cap = (int(nv_max / n_shards * 1.08) + 65536) // 4096 * 4096
    shards = [DevEngine(hip_lib, int(sm.counts[s]), cap, torch) for s in range(n_shards)]
    router = shards[0]
    g2l = torch.from_numpy(sm.local).to(dev)
    off = torch.zeros(n_shards + 1, dtype=torch.int32, device=dev)
    # the router engine's batch limit is its own shard's: route the batch in slices that fit (a stable partition
    # of consecutive slices, shard by shard, is the stable partition of the whole)
    for r, cols in enumerate(rounds):
        n = cols[0].shape[0]
        cols7 = list(cols) + [np.arange(n, dtype=np.int32)]  # the record's index travels as a seventh column
        d_in = [torch.from_numpy(c).to(dev) for c in cols7]
        per_shard = [[] for _ in range(n_shards)]
        torch.cuda.synchronize()
        for lo in range(0, n, cap):
            hi = min(n, lo + cap)
            d_out = [torch.empty(hi - lo, dtype=torch.int32, device=dev) for _ in cols7]
            router.e.route_dev(hi - lo, [t[lo:].data_ptr() for t in d_in], g2l.data_ptr(), G, n_shards,
                               [t.data_ptr() for t in d_out], off.data_ptr())
            router.e.sync()
            off_h = off.cpu().numpy()
            assert off_h[0] == 0 and off_h[-1] == hi - lo
            for s in range(n_shards):
                per_shard[s].append([t[int(off_h[s]):int(off_h[s + 1])] for t in d_out])
        routed = [[torch.cat([piece[k] for piece in per_shard[s]]) for k in range(7)] for s in range(n_shards)]
        torch.cuda.synchronize()
        # every engine on its own stream, no engine waits for another
        for s, sh in enumerate(shards):
            sh.propose_all()
            sh.accept_reply(int(routed[s][0].shape[0]), [t.data_ptr() for t in routed[s][:6]])
        ph_one, dh_one, st_one = ref[r]
        status = np.full(n, 255, np.uint8)
        merged = []
        for s, sh in enumerate(shards):
            ph, dec, st = sh.results()
            gl = sm.globals_of[s]
            for x, y, nm in zip(ph, ph_one, ("slot", "bnum", "bcoord", "median", "status")):
                assert (x == y[gl]).all(), f"round {r} shard {s}: propose {nm}"
            dec[:, 0] = sm.to_global(s, dec[:, 0])
            merged.append(dec)
            status[routed[s][6].cpu().numpy()] = st
        assert (status == st_one).all(), f"round {r}: per-vote status through the shards"
        merged = np.concatenate(merged)
        # a shard's stream is grouped by local index ascending = global index ascending (ShardMap numbers a shard's
        # groups in global order), so a stable sort by group is the merge the single engine's order asks for
        merged = merged[np.argsort(merged[:, 0], kind="stable")]
        assert merged.shape == dh_one.shape and (merged == dh_one).all(), f"round {r}: union of the decided streams"
    total = [0, 0, 0]
    for s, sh in enumerate(shards):
        gl = sm.globals_of[s]
        rows = sh.e.snapshot(np.arange(gl.shape[0], dtype=np.int32))[0]
        assert rows.tobytes() == rows_one[gl].tobytes(), f"shard {s}: HotRestoreInfo rows"
        total = [a + b for a, b in zip(total, sh.e.counters())]
        sh.close()
    assert tuple(total) == tuple(counters_one)
