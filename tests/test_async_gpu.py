"""The asynchronous host-pointer calls (gpx_*_batch_async + gpx_engine_wait, include/gpx.h): several calls in
flight - inputs of call N + 1 on their way in while call N's kernels run and call N - 1's outputs travel back -
must give exactly the answers of the synchronous calls (oracle), in submission order."""
import numpy as np
import pytest

from gigapaxos_amd import Engine, hri_create, streams, S_OK, D_DECISION
from gigapaxos_amd._abi import GpxError
from tests.parity_common import make_pair

pytestmark = pytest.mark.gpu


def _same(dh, do, what):
    a, b = dh.as_tuple_array(), do.as_tuple_array()
    assert a.shape == b.shape and (a == b).all(), what
    assert (dh.status == do.status).all(), what


@pytest.mark.parametrize("G,k", [(3000, 3), (300_000, 3), (100_000, 5)])
def test_async_rounds_match_oracle(hip_lib, oracle_lib, G, k):
    """Full rounds on one replica through the async calls, up to four in flight: propose -> its ACCEPTs ->
    votes (every other round without ballot columns: the common-ballot form) -> commits; the oracle goes through
    the synchronous calls."""
    members = list(range(100, 100 + k))
    eh, eo = make_pair(hip_lib, oracle_lib, 100, G, k, 8, max_batch=G * k + G * k // 40 + 4096)
    mem = np.tile(np.array(members, np.int32), (G, 1))
    for e in (eh, eo):
        assert (e.create_groups(np.arange(G), mem, k, hri_create(G, k, 100)) == S_OK).all()
    g = np.arange(G, dtype=np.int32)
    for r in range(5):
        # two calls in flight: the proposal and (its results are known to the test from the oracle) the votes
        po = eo.propose(g)
        tp = eh.propose_async(g)
        cols = streams.vote_round(G, members, r, 100, config_id=4, mix=(r == 3))
        common = r % 2 == 0 and r != 3
        tv = (eh.accept_reply_async(cols[0], None, None, cols[3], cols[4], cols[5], common_ballot=(0, 100)) if common
              else eh.accept_reply_async(*cols))
        do = eo.accept_reply(*cols)
        # ... and two more behind them: the round's ACCEPTs and commits on the acceptor side of the same engine
        ta = eh.accept_async(g, po[1], po[2], po[0], po[3])
        tc = eh.commit_async(do.gidx, do.bnum, do.bcoord, do.slot, do.median_cp, np.full(do.gidx.shape[0], 1, np.uint8))
        with pytest.raises(GpxError):            # a fifth call: GPX_EBUSY until a ticket is waited for
            eh.propose_async(g[:4])
        ph = tp.wait()
        for x, y in zip(ph, po):
            assert (x == y).all()
        _same(tv.wait(), do, f"round {r} votes")
        (ra, xa), (rb, xb) = ta.wait(), eo.accept(g, po[1], po[2], po[0], po[3])
        for x, y in zip(ra, rb):
            assert (x == y).all()
        assert (xa.as_tuple_array() == xb.as_tuple_array()).all()
        (sa, ca), (sb, cb) = tc.wait(), eo.commit(do.gidx, do.bnum, do.bcoord, do.slot, do.median_cp,
                                                   np.full(do.gidx.shape[0], 1, np.uint8))
        assert (sa == sb).all() and (ca.as_tuple_array() == cb.as_tuple_array()).all()
        with pytest.raises(GpxError):            # a ticket is good for one wait
            tc.wait()
    assert eh.snapshot(g)[0].tobytes() == eo.snapshot(g)[0].tobytes()
    assert eh.counters() == eo.counters()
    eh.close()
    eo.close()


@pytest.mark.parametrize("pin_outputs", [False, True])
def test_async_pipeline_of_vote_batches(hip_lib, oracle_lib, pin_outputs):
    """The bench's shape: a stream of (propose, votes) steps kept two steps deep from pinned buffers; with the
    outputs pinned too the engine writes the decisions into them itself (k_copy_out: exactly n_out entries, no
    host round trip for the count)."""
    G, k, R = 200_000, 3, 6
    members = [100, 101, 102]
    eh, eo = make_pair(hip_lib, oracle_lib, 100, G, k, 8, max_batch=G * k + 4096)
    mem = np.tile(np.array(members, np.int32), (G, 1))
    for e in (eh, eo):
        assert (e.create_groups(np.arange(G), mem, k, hri_create(G, k, 100)) == S_OK).all()
    g = np.arange(G, dtype=np.int32)
    rounds = [streams.vote_round(G, members, r, 100) for r in range(R)]
    pinned = [c for cols in rounds for c in cols] + [g]
    eh.host_register(*pinned)
    pend = []
    got = []
    for r in range(R):
        pend.append((eh.propose_async(g, pin_outputs=pin_outputs),
                     eh.accept_reply_async(*rounds[r], pin_outputs=pin_outputs)))
        if len(pend) == 2:
            tp, tv = pend.pop(0)
            tp.wait()
            got.append(tv.wait())
    for tp, tv in pend:
        tp.wait()
        got.append(tv.wait())
    eh.host_unregister(*pinned)
    for r in range(R):
        eo.propose(g)
        do = eo.accept_reply(*rounds[r])
        _same(got[r], do, f"round {r}")
        assert got[r].gidx.shape[0] == G and (got[r].kind == D_DECISION).all()
    assert eh.snapshot(g)[0].tobytes() == eo.snapshot(g)[0].tobytes()
    eh.close()
    eo.close()


def test_unregister_with_a_ticket_in_flight_then_pageable_copies(hip_lib, oracle_lib):
    """gpx_host_unregister while an asynchronous call still writes through the block's device
    mapping (k_copy_out on the set's copy-out stream) - the call must drain the engine's streams first, so that the
    outputs are complete when it returns, the ticket can still be waited for, and the freed range can be used for
    large pageable copies at once (the SIGABRT of profiles/r03_pytest_gpu_abort_incident.log was torch's first big
    pageable H2D copy after such traffic).  Then gpx_engine_destroy with registered blocks and a ticket nobody
    waited for: it unregisters what is left itself."""
    import torch
    G, k = 400_000, 3
    members = [100, 101, 102]
    eh, eo = make_pair(hip_lib, oracle_lib, 100, G, k, 8, max_batch=G * k + 4096)
    mem = np.tile(np.array(members, np.int32), (G, 1))
    for e in (eh, eo):
        assert (e.create_groups(np.arange(G), mem, k, hri_create(G, k, 100)) == S_OK).all()
    g = np.arange(G, dtype=np.int32)
    for r in range(3):
        cols = streams.vote_round(G, members, r, 100)
        eo.propose(g)
        do = eo.accept_reply(*cols)
        eh.propose_async(g).wait()
        eh.host_register(*cols)
        tv = eh.accept_reply_async(*cols, pin_outputs=True)     # outputs registered: k_copy_out writes them
        # take the INPUT registrations away at once, the call still in flight ...
        eh.host_unregister(*cols)
        # ... then big pageable copies through the same process (what test_route_gpu.py does first)
        big = torch.from_numpy(np.arange(8_000_000, dtype=np.int64))
        assert int(big.cuda().sum().item()) == 8_000_000 * (8_000_000 - 1) // 2
        _same(tv.wait(), do, f"round {r}")                       # the ticket is still good; finish() unregisters the outputs
    # a ticket nobody waits for and blocks still registered at destroy
    cols = streams.vote_round(G, members, 3, 100)
    eh.propose_async(g).wait()
    eh.host_register(*cols)
    pending = eh.accept_reply_async(*cols)                       # (keeps the call's buffers alive until the engine is gone)
    eh.close()                                                   # drains, unregisters cols itself
    del pending
    big = torch.from_numpy(np.arange(4_000_000, dtype=np.int64))
    assert int(big.cuda().sum().item()) == 4_000_000 * (4_000_000 - 1) // 2
    t = torch.from_numpy(cols[0])                                # the range is ordinary pageable memory again
    assert int(t.cuda().sum().item()) == int(cols[0].astype(np.int64).sum())
    eo.close()
