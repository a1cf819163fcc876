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
