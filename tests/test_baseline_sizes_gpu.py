"""BASELINE.md section 4 at the sizes it states, on the HIP engine against the oracle:
config #1 = `tests/loopback_1_group` semantics, 10,000 requests through 1 group on 3 replicas (the reference's
TESTPaxosClient sends them one at a time: one slot per request); config #2 = 100 rounds x 10,000 proposals through
10,000 groups on 3 replicas, whole pipeline (propose -> ACCEPT x3 -> replies -> decision -> BATCHED_COMMIT x3 ->
in-order execution).  Every round's decided stream is compared, then every replica's execution log, every
HotRestoreInfo row and sampled full state dumps.  (tests/test_parity_gpu.py keeps short versions of both that run
under each accept-reply path.)"""
import numpy as np
import pytest

from gigapaxos_amd import D_DECISION
from gigapaxos_amd.loopback import LoopbackCluster
from tests.parity_common import assert_same_state

pytestmark = pytest.mark.gpu

NODES = [100, 101, 102]


def test_config1_loopback_one_group_10000_requests(hip_lib, oracle_lib):
    n = 10_000
    ch = LoopbackCluster(hip_lib, NODES, 1, window=8, max_batch=1024)
    co = LoopbackCluster(oracle_lib, NODES, 1, window=8)
    for r in range(n):
        dh, do = ch.round([0]), co.round([0])
        assert dh.tolist() == do.tolist(), f"request {r}"
        assert dh[0, :4].tolist() == [0, r + 1, 0, 100] and dh[0, 5] == D_DECISION
    for nid in NODES:
        ex = ch.executed(nid)
        assert ex.tolist() == co.executed(nid).tolist()
        slots = np.concatenate([np.arange(f, f + cnt) for _, f, cnt in ex])
        assert slots.tolist() == list(range(1, n + 1))  # every request executed once, in slot order, on every replica
        assert ch.engines[nid].snapshot([0])[0].tobytes() == co.engines[nid].snapshot([0])[0].tobytes()
        assert_same_state(ch.engines[nid], co.engines[nid], [0])
        assert ch.engines[nid].counters() == co.engines[nid].counters()
    ch.close()
    co.close()


def test_config2_10k_groups_100_rounds_full_pipeline(hip_lib, oracle_lib):
    G, R = 10_000, 100
    rng = np.random.default_rng(2)
    coord = rng.choice(NODES, size=G).astype(np.int32)  # as roundRobinCoordinator(name) spreads them (PISM:2251-2256)
    ch = LoopbackCluster(hip_lib, NODES, G, window=8, max_batch=1 << 16, coordinator=coord)
    co = LoopbackCluster(oracle_lib, NODES, G, window=8, coordinator=coord)
    for r in range(R):
        groups = rng.permutation(G).astype(np.int32)
        dh, do = ch.round(groups), co.round(groups)
        assert dh.shape[0] == G and dh.tolist() == do.tolist(), f"round {r}"
    for nid in NODES:
        assert ch.executed(nid).tolist() == co.executed(nid).tolist()
        eh, eo = ch.engines[nid], co.engines[nid]
        sh, so = eh.snapshot(np.arange(G))[0], eo.snapshot(np.arange(G))[0]
        assert sh.tobytes() == so.tobytes()
        assert (sh["acc_slot"] == R + 1).all()
        assert_same_state(eh, eo, rng.integers(0, G, 40))
        assert eh.counters() == eo.counters()
    ch.close()
    co.close()
