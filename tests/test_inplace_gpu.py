"""The accept-reply call's outputs written IN PLACE by the per-bucket kernel of the tiled front end (PlaceCols,
gigapaxos_amd/csrc/gpx_ar16.hip.h): a bucket predicts where its decisions go from the votes before it
(every D votes one output, D kept on the device and learnt from the last call that had to be compacted), writes there
when its count is the predicted one, and k_emit_dec16 compacts the staging only when some bucket's was not.  The results
must be the oracle's either way; gpx_engine_path_counters says which way a call went.  Follows
PISM.handleBatchedAcceptReply (PaxosInstanceStateMachine.java:1370-1419) -> PCS.handleAcceptReplyMyBallot
(PaxosCoordinatorState.java:597-640)."""
import numpy as np
import pytest

from gigapaxos_amd import hri_create, streams, S_OK
from tests.parity_common import make_pair
from tests.test_fullsize_gpu import _same

pytestmark = pytest.mark.gpu


def _pair(hip_lib, oracle_lib, G, k):
    members = list(range(100, 100 + k))
    eh, eo = make_pair(hip_lib, oracle_lib, 100, G, k, 8, max_batch=G * k + G * k // 25 + 4096)
    mem = np.tile(np.array(members, np.int32), (G, 1))
    for e in (eh, eo):
        assert (e.create_groups(np.arange(G), mem, k, hri_create(G, k, 100)) == S_OK).all()
    return eh, eo, members


@pytest.mark.parametrize("G,k", [(250_123, 3), (125_000, 5)])
def test_steady_state_in_place_and_the_ratio_is_learnt(hip_lib, oracle_lib, G, k):
    """Rounds of one engine: every replica answers (in place from the first call: D starts at the replica count); one
    acceptor's votes lost (every bucket's count differs from its span: compacted, and D becomes votes / outputs);
    the same again (in place with the learnt D); every replica again (compacted once, then in place); the adversarial
    mix (compacted).  The table's last bucket is partly filled."""
    eh, eo, members = _pair(hip_lib, oracle_lib, G, k)
    g = np.arange(G, dtype=np.int32)
    plan = ["all", "lost", "lost", "all", "all", "mix", "all"]
    want_placed = [True, False, True, False, True, False, None]
    placed = compacted = 0
    for r, (what, wp) in enumerate(zip(plan, want_placed)):
        for x, y in zip(eh.propose(g), eo.propose(g)):
            assert (x == y).all()
        cols = streams.vote_round(G, members, r, 100, config_id=3 if k == 3 else 4, mix=what == "mix")
        if what == "lost":
            keep = cols[4] != members[-1]
            cols = [np.ascontiguousarray(c[keep]) for c in cols]
        _same(eh.accept_reply(*cols), eo.accept_reply(*cols), f"round {r} ({what})")
        p, c = eh.path_counters()
        assert p + c == r + 1
        if wp is not None:
            assert (p == placed + 1) == wp, f"round {r} ({what}): in place {p}, compacted {c}"
        placed, compacted = p, c
    assert eh.snapshot(g)[0].tobytes() == eo.snapshot(g)[0].tobytes()
    assert eh.counters() == eo.counters()
    eh.close()
    eo.close()


def test_staging_only_form_gives_the_same(hip_lib, oracle_lib, monkeypatch):
    """GPX_AR_INPLACE=0: the per-bucket kernel stages only and k_emit_dec16 always compacts (round 5's form)."""
    monkeypatch.setenv("GPX_AR_INPLACE", "0")
    G, k = 200_000, 3
    eh, eo, members = _pair(hip_lib, oracle_lib, G, k)
    g = np.arange(G, dtype=np.int32)
    for r in range(2):
        for x, y in zip(eh.propose(g), eo.propose(g)):
            assert (x == y).all()
        cols = streams.vote_round(G, members, r, 100, config_id=3, mix=r == 1)
        _same(eh.accept_reply(*cols), eo.accept_reply(*cols), f"round {r}")
    assert eh.path_counters() == (0, 0)
    eh.close()
    eo.close()


def test_nothing_is_written_beyond_the_columns(hip_lib, oracle_lib):
    """The device entry point with output columns of exactly n entries inside larger allocations: the in-place stores
    of a steady-state call and the predicted stores of a call that turns out not to be one stay below n; what lies at
    and beyond *n_out is unspecified (include/gpx.h), what lies beyond n is untouched."""
    import torch
    G, k = 150_000, 3
    eh, eo, members = _pair(hip_lib, oracle_lib, G, k)
    g = np.arange(G, dtype=np.int32)
    GUARD = 4096
    for r, what in enumerate(["all", "lost", "mix"]):
        for x, y in zip(eh.propose(g), eo.propose(g)):
            assert (x == y).all()
        cols = streams.vote_round(G, members, r, 100, config_id=3, mix=what == "mix")
        if what == "lost":      # D = 3 on the device, two votes per group: every bucket decides MORE than its span
            keep = cols[4] != members[0]
            cols = [np.ascontiguousarray(c[keep]) for c in cols]
        n = cols[0].shape[0]
        dc = [torch.from_numpy(c).cuda() for c in cols]
        d = [torch.full((n + GUARD,), -77, dtype=torch.int32, device="cuda") for _ in range(5)] + \
            [torch.full((n + GUARD,), 201, dtype=torch.uint8, device="cuda")]
        no, st = torch.zeros(1, dtype=torch.int32, device="cuda"), torch.zeros(n, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        eh.call_dev("accept_reply_batch", n, *[t.data_ptr() for t in dc], *[t.data_ptr() for t in d], no.data_ptr(), st.data_ptr())
        eh.sync()
        do = eo.accept_reply(*cols)
        m = int(no.item())
        got = np.stack([t[:m].cpu().numpy().astype(np.int32) for t in d], axis=1)
        assert got.shape == do.as_tuple_array().shape and (got == do.as_tuple_array()).all(), what
        assert (st.cpu().numpy() == do.status).all()
        for t, fill in zip(d, [-77] * 5 + [201]):
            assert (t[n:] == fill).all(), f"{what}: a store beyond the column's n entries"
    eh.close()
    eo.close()
