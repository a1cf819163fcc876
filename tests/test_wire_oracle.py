"""Pins the wire-codec ORACLE (oracle/gpx_wire_oracle.inc) against what the reference itself fixes
about these formats: BatchedAcceptReply.main's coalesced "pid1" packet
(paxospackets/BatchedAcceptReply.java:220-240), the SIZEOF_* constants, the TreeMap / TreeSet
ordering of slot lists, the constructors' failure modes, PaxosManager's name / version demux and
PaxosPacketBatcher's (group, ballot) coalescing of decisions.  CPU only."""
import struct

import numpy as np
import pytest

from gigapaxos_amd import Engine, hri_create, S_OK, S_EXISTS, S_NOGROUP, D_DECISION, D_PREEMPTED
from gigapaxos_amd import wire as W
from gigapaxos_amd._abi import Decisions


def make_engine(lib, G=64, k=3, my_id=100, named=16, version=0):
    e = Engine(lib, my_id, G, kmax=k, window=8)
    we = W.WireEngine(e)
    members = np.tile(np.arange(100, 100 + k, dtype=np.int32), (named, 1))
    rows = hri_create(named, k, my_id)
    rows["version"] = version
    assert (e.create_groups(np.arange(named), members, k, rows) == S_OK).all()
    names = [b"pid%d" % i for i in range(named)]
    assert (we.bind(names, np.arange(named)) == S_OK).all()
    return e, we, names


def test_sizeof_constants_and_layout():
    """SIZEOF_PAXOSPACKET_FIXED = 13 (PaxosPacket.java:459), SIZEOF_ACCEPTREPLY = 29
    (AcceptReplyPacket.java:121-127), batched: + 4 + 12 n (BatchedAcceptReply.java:139-147);
    SIZEOF_BATCHEDCOMMIT_FIXED = 12 (BatchedCommit.java:181-183) + 4 (n + 1 + g + 1);
    SIZEOF_REQUEST_FIXED = 55 (RequestPacket.java:779-797); accept tail 4 + 14 + 4
    (ProposalPacket.java:82, PValuePacket.java:81, AcceptPacket.java:93)."""
    bar = W.batched_accept_reply(b"pid1", 0, 23, 0, 234, -1, [1, 2])
    assert len(bar) == 13 + 4 + 29 + 4 + 12 * 2
    assert bar[:13] == struct.pack(">iiib", 90, 34, 0, 4) and bar[13:17] == b"pid1"
    bc = W.batched_commit(b"pid1", 0, 0, 234, 7, [3, 4, 5], [101, 102])
    assert len(bc) == 13 + 4 + 12 + 4 * (3 + 1 + 2 + 1)
    rq = W.request(b"pid1", 0, 77, b"x" * 64)
    assert len(rq) == 13 + 4 + 55 + 64
    ac = W.accept(b"pid1", 0, 77, 9, 0, 234, 5, 100, b"x" * 64)
    assert len(ac) == len(rq) + 4 + 14 + 4
    assert W.java_string_hash(b"hello") == 99162322


def test_batched_accept_reply_main_fixture(oracle_lib):
    """BatchedAcceptReply.main: AcceptReplyPacket(23, Ballot(0,234), slot 1, maxCP -1) and slot 2 of
    "pid1" coalesced into one BatchedAcceptReply; bytes -> packet keeps both slots in order."""
    e, we, names = make_engine(oracle_lib)
    bar = W.batched_accept_reply(b"pid1", 0, 23, 0, 234, -1, [1, 2])
    d = we.decode([bar])
    assert d.f_status.tolist() == [W.W_OK] and d.f_gidx.tolist() == [1] and d.f_type.tolist() == [34]
    v = d.votes
    assert v["gidx"].tolist() == [1, 1] and v["slot"].tolist() == [1, 2]
    assert v["bnum"].tolist() == [0, 0] and v["bcoord"].tolist() == [234, 234]
    assert v["acceptor"].tolist() == [23, 23] and v["max_cp"].tolist() == [-1, -1]
    assert d.counts["n_votes"] == 2 and d.counts["n_bad_frames"] == 0


def test_slot_lists_are_sorted_sets(oracle_lib):
    """TreeMap<Integer,Long>.put / TreeSet.add: signed ascending order, duplicates collapse."""
    e, we, names = make_engine(oracle_lib)
    bar = W.batched_accept_reply(b"pid2", 0, 101, 0, 100, 4, [7, -3, 7, 5, 2**31 - 1, -2**31])
    bc = W.batched_commit(b"pid3", 0, 0, 100, 1, [9, 8, 9, 8, 1], [101, 102])
    d = we.decode([bar, bc])
    assert d.votes["slot"].tolist() == [-2**31, -3, 5, 7, 2**31 - 1]
    assert d.votes["frame"].tolist() == [0] * 5
    assert d.commits["slot"].tolist() == [1, 8, 9] and d.commits["gidx"].tolist() == [3, 3, 3]
    assert d.commits["kind"].tolist() == [0, 0, 0] and d.commits["median_cp"].tolist() == [1, 1, 1]
    # non-ascending lists longer than the engine limit are refused (gpx_wire.h)
    big = list(range(W.W_MAX_UNSORTED + 1, 0, -1))
    d = we.decode([W.batched_commit(b"pid3", 0, 0, 100, 1, big, []),
                   W.batched_commit(b"pid3", 0, 0, 100, 1, sorted(big), [])])
    assert d.f_status.tolist() == [W.W_MALFORMED, W.W_OK]
    assert d.commits["slot"].tolist() == sorted(big)


def test_demux_name_and_version(oracle_lib):
    """PaxosManager.handlePaxosPacket: unknown paxosID or version mismatch -> dropped."""
    e, we, names = make_engine(oracle_lib, version=3)
    frames = [
        W.batched_commit(b"pid1", 3, 0, 100, 0, [1], [101, 102]),
        W.batched_commit(b"nosuch", 3, 0, 100, 0, [1], [101, 102]),
        W.batched_commit(b"pid1", 2, 0, 100, 0, [2], [101, 102]),
        W.batched_commit(b"pid15", 3, 0, 100, 0, [3], [101, 102]),
    ]
    d = we.decode(frames)
    assert d.f_status.tolist() == [W.W_OK, W.W_NOGROUP, W.W_VERSION, W.W_OK]
    assert d.f_gidx.tolist() == [1, -1, 1, 15]
    assert d.commits["slot"].tolist() == [1, 3] and d.commits["frame"].tolist() == [0, 3]
    assert d.counts["n_bad_frames"] == 2
    # unbind: the name disappears, the row can be bound to another name
    assert we.unbind([15]).tolist() == [S_OK]
    assert we.lookup([b"pid15", b"pid14"]).tolist() == [-1, 14]
    assert we.bind([b"pid14"], [15]).tolist() == [S_EXISTS]  # name taken
    assert we.bind([b"other"], [14]).tolist() == [S_EXISTS]  # row taken
    assert we.bind([b"other", b""], [15, 20]).tolist() == [S_OK, S_NOGROUP]
    d = we.decode([W.batched_commit(b"other", 3, 0, 100, 0, [4], [])])
    assert d.f_gidx.tolist() == [15] and d.f_status.tolist() == [W.W_OK]


def test_constructor_failure_modes(oracle_lib):
    """BufferUnderflowException / NegativeArraySizeException / unknown type ints drop the packet."""
    e, we, names = make_engine(oracle_lib)
    good_bar = W.batched_accept_reply(b"pid1", 0, 101, 0, 100, 0, [1, 2, 3])
    good_bc = W.batched_commit(b"pid1", 0, 0, 100, 0, [1, 2], [101, 102])
    good_acc = W.accept(b"pid1", 0, 5, 1, 0, 100, 0, 100, b"v" * 10)
    neg_n = bytearray(good_bar)
    neg_n[13 + 4 + 29:13 + 4 + 33] = struct.pack(">i", -5)
    frames = [
        good_bar[:-1],                                   # truncated slot list
        bytes(neg_n),                                    # numSlots < 0: loop never runs, 0 votes, OK
        good_bc[:-4],                                    # truncated group list
        good_bc + b"\x00" * 7,                           # trailing bytes are ignored
        struct.pack(">ii", 91, 34) + good_bar[8:],       # not a PAXOS_PACKET
        struct.pack(">ii", 90, 12) + good_bar[8:],       # unknown PaxosPacketType int
        struct.pack(">ii", 90, 2) + good_bar[8:],        # PREPARE: known type, not byteified
        good_bar[:12] + b"\xff" + good_bar[13:],         # paxosIDLength = -1
        good_acc[:-3],                                   # accept tail cut
        good_acc,
        b"\x00\x00",                                     # shorter than the type ints
        W.batched_accept_reply(b"", 0, 101, 0, 100, 0, [1]),  # null paxosID: BAR ctor NPE
        W.batched_commit(b"", 0, 0, 100, 0, [1], []),         # null paxosID: no instance
    ]
    d = we.decode(frames)
    M, OK, U, NG = W.W_MALFORMED, W.W_OK, W.W_UNSUPPORTED, W.W_NOGROUP
    assert d.f_status.tolist() == [M, OK, M, OK, M, M, U, M, M, OK, M, M, NG]
    assert d.f_type.tolist() == [-1, 34, -1, 35, -1, -1, 2, -1, -1, 3, -1, -1, 35]
    assert d.counts == {"n_votes": 0, "n_commits": 2, "n_accepts": 1, "n_requests": 0, "n_bad_frames": 10}


def test_request_and_accept_fields(oracle_lib):
    """RequestPacket(ByteBuffer) walk incl. batched sub-requests (RequestPacket.java:956-1020);
    isStopRequest = own flag or any batched request's (:1069-1080); accept tail."""
    e, we, names = make_engine(oracle_lib)
    sub = [W.request(b"pid4", 0, 1000 + i, b"s" * i, stop=(i == 2)) for i in range(4)]
    deep = W.request(b"pid4", 0, 2000, b"", batched=[W.request(b"pid4", 0, 2001, b"", stop=True)])
    frames = [
        W.request(b"pid4", 0, 11, b"hello"),
        W.request(b"pid4", 0, 12, b"hello", stop=True),
        W.request(b"pid4", 0, 13, b"hello", batched=sub[:2]),
        W.request(b"pid4", 0, 14, b"hello", batched=sub),
        W.request(b"pid4", 0, 15, b"", batched=[deep]),
        W.request(b"pid4", 0, -2, b"", digest=b"d" * 20, response=b"resp"),
        W.accept(b"pid5", 0, 2**40 + 7, -9, 3, 102, 6, 102, b"v" * 64, batched=sub),
        W.accept(b"pid5", 0, 16, 10, 0, 100, -1, 100, b"", recovery=True),
    ]
    # a nested element whose own length field lies: the element constructor underflows
    bad_nested = W.request(b"pid4", 0, 17, b"", batched=[sub[1][:-2]])
    frames.append(bad_nested)
    d = we.decode(frames)
    assert d.f_status.tolist() == [W.W_OK] * 8 + [W.W_MALFORMED]
    r = d.requests
    assert r["gidx"].tolist() == [4] * 6 and r["req_id"].tolist() == [11, 12, 13, 14, 15, -2]
    assert r["is_stop"].tolist() == [0, 1, 0, 1, 1, 0] and r["frame"].tolist() == [0, 1, 2, 3, 4, 5]
    a = d.accepts
    assert a["gidx"].tolist() == [5, 5] and a["req_id"].tolist() == [2**40 + 7, 16]
    assert a["slot"].tolist() == [-9, 10] and a["bnum"].tolist() == [3, 0] and a["bcoord"].tolist() == [102, 100]
    assert a["median_cp"].tolist() == [6, -1] and a["sender"].tolist() == [102, 100]
    assert a["flags"].tolist() == [1, 0] and a["frame"].tolist() == [6, 7]


def test_capacity_is_reported_not_overrun(oracle_lib):
    e, we, names = make_engine(oracle_lib)
    frames = [W.batched_accept_reply(b"pid%d" % i, 0, 101, 0, 100, 0, [1, 2, 3]) for i in range(4)]
    d = we.decode(frames, cap_votes=7)
    assert d.f_status.tolist() == [W.W_OK, W.W_OK, W.W_CAPACITY, W.W_CAPACITY]
    assert d.counts["n_votes"] == 12 and d.votes["gidx"].tolist()[:6] == [0, 0, 0, 1, 1, 1]


def _decisions(rows):
    a = np.array(rows, np.int32).reshape(-1, 6)
    return Decisions(a[:, 0].copy(), a[:, 1].copy(), a[:, 2].copy(), a[:, 3].copy(), a[:, 4].copy(),
                     a[:, 5].astype(np.uint8), np.zeros(0, np.uint8))


def test_pack_commits_batcher_semantics(oracle_lib):
    """PaxosPacketBatcher: all decisions of one (paxosID, ballot) coalesce into one BatchedCommit:
    slots a TreeSet, median folded with `b - cur > 0`, group = members minus self (TreeSet);
    PREEMPTED rows never coalesce; BatchedCommit.toBytes layout."""
    e, we, names = make_engine(oracle_lib, k=3, my_id=101)
    dec = _decisions([
        (2, 5, 0, 101, 3, D_DECISION), (2, 4, 0, 101, 4, D_DECISION), (2, 9, 0, 101, 2, D_PREEMPTED),
        (2, 6, 0, 101, 1, D_DECISION),
        (7, 1, 1, 101, -1, D_DECISION), (7, 2, 2, 101, 0, D_DECISION), (7, 3, 1, 101, 5, D_DECISION),
        (9, 8, 0, 101, 0, D_PREEMPTED),
    ])
    frames, fg, nbytes = we.pack_commits(dec)
    assert fg.tolist() == [2, 7, 7]
    assert frames[0] == W.batched_commit(b"pid2", 0, 0, 101, 4, [4, 5, 6], [100, 102])
    assert frames[1] == W.batched_commit(b"pid7", 0, 1, 101, 5, [1, 3], [100, 102])
    assert frames[2] == W.batched_commit(b"pid7", 0, 2, 101, 0, [2], [100, 102])
    assert nbytes == sum((len(f) + 3) // 4 * 4 for f in frames)
    # and back: what another replica's decoder makes of them
    e2, we2, _ = make_engine(oracle_lib, k=3, my_id=100)
    d = we2.decode(frames)
    assert d.f_status.tolist() == [W.W_OK] * 3
    assert d.commits["gidx"].tolist() == [2, 2, 2, 7, 7, 7]
    assert d.commits["slot"].tolist() == [4, 5, 6, 1, 3, 2]
    assert d.commits["median_cp"].tolist() == [4, 4, 4, 5, 5, 0]
    assert d.commits["bnum"].tolist() == [0, 0, 0, 1, 1, 2]


def test_rows_allocator(oracle_lib):
    e = Engine(oracle_lib, 100, 8, kmax=3, window=8)
    we = W.WireEngine(e)
    a = we.rows_alloc(5)
    assert a.tolist() == [0, 1, 2, 3, 4]
    we.rows_free([3, 1])
    assert we.rows_alloc(4).tolist() == [1, 3, 5, 6]
    with pytest.raises(Exception):
        we.rows_alloc(2)
    assert we.rows_alloc(1).tolist() == [7]


def test_pack_accept_replies_batcher_semantics(oracle_lib):
    """PaxosPacketBatcher.enqueueImpl(AcceptReplyPacket): one BatchedAcceptReply per (paxosID,
    ballot) keeps the FIRST reply's slot / maxCheckpointedSlot / requestID in its fixed part and
    TreeMap-merges the slots (a repeated slot keeps the last request id); replies whose ballot
    coordinator is not the ACCEPT's sender are not coalescable; dropped accepts have no reply."""
    e, we, names = make_engine(oracle_lib, k=3, my_id=101)
    #        gidx slot bnum bcoord maxcp status sender reqid
    rows = [(3, 7, 0, 100, 5, 0, 100, 70),
            (5, 1, 0, 102, 0, 0, 102, 10),
            (3, 6, 0, 100, 9, 0, 100, 60),
            (3, 7, 0, 100, 9, 0, 100, 71),   # same slot again: request id 71 wins, maxCP stays 5
            (3, 8, 1, 102, 9, 0, 100, 80),   # acceptor already promised (1,102): NACK, not coalescable
            (3, 9, 1, 102, 9, 0, 102, 90),   # a second ballot of group 3
            (5, 2, 0, 102, 0, 2, 102, 20),   # status STOPPED: no reply at all
            (40, 1, 0, 100, 0, 0, 100, 1),   # row without a name
            (-1, 1, 0, 100, 0, 1, 100, 1)]   # unknown group: dropped by the accept call
    a = np.array(rows, np.int64)
    frames, fg, fd, ub, nbytes = we.pack_accept_replies(a[:, 0], a[:, 1], a[:, 2], a[:, 3], a[:, 4],
                                                        a[:, 5].astype(np.uint8), sender=a[:, 6], req_id=a[:, 7])
    assert fg.tolist() == [3, 3, 5] and fd.tolist() == [100, 102, 102]
    assert frames[0] == W.batched_accept_reply(b"pid3", 0, 101, 0, 100, 5, [6, 7], [60, 71]) \
        .replace(struct.pack(">iiq", 6, 5, 60), struct.pack(">iiq", 7, 5, 70), 1)
    assert frames[1] == W.batched_accept_reply(b"pid3", 0, 101, 1, 102, 9, [9], [90])
    assert frames[2] == W.batched_accept_reply(b"pid5", 0, 101, 0, 102, 0, [1], [10])
    assert ub.tolist() == [0, 0, 0, 0, 1, 0, 0, 1, 0]
    assert nbytes == sum((len(f) + 3) // 4 * 4 for f in frames)
    # the coordinator's decoder turns them back into votes
    e2, we2, _ = make_engine(oracle_lib, k=3, my_id=100)
    d = we2.decode(frames[:1])
    assert d.votes["slot"].tolist() == [6, 7] and d.votes["max_cp"].tolist() == [5, 5]
    assert d.votes["acceptor"].tolist() == [101, 101]


def test_frame_level_cluster_executes_in_order(oracle_lib):
    """BASELINE config #1/#2 shape with every hop crossing the wire formats (tests/wire_cluster.py):
    3 replicas, coordinators spread over them, every replica executes every group's slots 1..R in
    order."""
    from tests.wire_cluster import WireCluster
    G, R = 60, 5
    names = [b"svc-%d" % g for g in range(G)]
    coord = np.array([100 + g % 3 for g in range(G)], np.int32)
    c = WireCluster(oracle_lib, [100, 101, 102], names, coord)
    rng = np.random.default_rng(3)
    for r in range(R):
        dec = c.round(rng.permutation(G), r)
        assert sum(d.gidx.shape[0] for d in dec.values()) == G
    for nid in (100, 101, 102):
        ex = c.executed(nid)
        for g in range(G):
            runs = ex[ex[:, 0] == g]
            slots = np.concatenate([np.arange(f, f + n) for _, f, n in runs])
            assert slots.tolist() == list(range(1, R + 1))
    c.close()


def test_pack_accept_replies_beyond_the_per_pass_limits(oracle_lib):
    """The reference's batcher maps are unbounded (PaxosPacketBatcher.java:121-137); the engine packs
    256 replies / 4 ballots of a group per pass and runs further passes over the rest, so NOTHING is
    left unbatched for that reason: 600 replies of one ballot leave as frames of 256 + 256 + 88 slots,
    six ballots as 4 + 2 frames."""
    e, we, names = make_engine(oracle_lib, k=3, my_id=101)
    n = 600
    g = np.full(n, 3, np.int32)
    slot = np.arange(1, n + 1, dtype=np.int32)
    z = np.zeros(n, np.int32)
    frames, fg, fd, ub, _ = we.pack_accept_replies(g, slot, z, z + 100, z, np.zeros(n, np.uint8), z + 100)
    assert not ub.any() and fg.tolist() == [3, 3, 3] and fd.tolist() == [100, 100, 100]
    sizes = []
    for f in frames:
        hdr = 13 + f[12]
        sizes.append(struct.unpack(">i", f[hdr + 29:hdr + 33])[0])
        first = struct.unpack(">i", f[hdr + 33:hdr + 37])[0]
        assert struct.unpack(">i", f[hdr + 12:hdr + 16])[0] == first   # the fixed part's slot = the pass's first reply
    assert sizes == [256, 256, 88]
    bn = np.arange(6, dtype=np.int32)
    frames, fg, fd, ub, _ = we.pack_accept_replies(np.full(6, 3, np.int32), np.arange(1, 7), bn, np.full(6, 100), z[:6],
                                                   np.zeros(6, np.uint8), np.full(6, 100))
    assert not ub.any() and len(frames) == 6
    got = [struct.unpack(">i", f[13 + f[12] + 4:13 + f[12] + 8])[0] for f in frames]
    assert got == [0, 1, 2, 3, 4, 5]


def test_plan_send_dequeue_bound_and_cross_group_batching(oracle_lib):
    """PaxosPacketBatcher.dequeueImpl's payload bound (:182-209: `while (lengthEstimate < MAX)`, test
    before add) and process() -> batch() (:268-303: more than MIN_PP_BATCH_SIZE = 3 tasks are regrouped by
    recipient set in first-appearance order), worked out by hand from the Java - and the engine
    library's own implementation of the same function (pure host code: runs without a GPU)."""
    import __graft_entry__ as ge
    from gigapaxos_amd._abi import GpxLib

    ge.build()
    libs = [oracle_lib, GpxLib(ge.HIP_SO, "gpx_", device_api=True)]
    for lib in libs:
        # five frames of estimate 10 under a bound of 25: 0 < 25, 10 < 25, 20 < 25 -> three leave (30 >= 25
        # stops the loop AFTER the third); the next dequeue takes the other two.  3 tasks is not MORE
        # than 3: sent one by one.
        burst, env, pos, nb = W.plan_send(lib, [10] * 5, [7, 7, 8, 7, 8], max_payload=25)
        assert nb == 2 and burst.tolist() == [0, 0, 0, 1, 1]
        assert env.tolist() == [-1] * 5 and pos.tolist() == [0] * 5
        # six frames in one dequeue (> 3 tasks): LinkedHashMap order of the recipient sets 5, 9, 2
        burst, env, pos, nb = W.plan_send(lib, [1] * 6, [5, 9, 5, 2, 9, 5], max_payload=100)
        assert nb == 1 and burst.tolist() == [0] * 6
        assert env.tolist() == [0, 1, 0, 2, 1, 0] and pos.tolist() == [0, 0, 1, 0, 1, 2]
        # BATCH_ACROSS_GROUPS off: never regrouped; a single frame above the bound still leaves alone
        burst, env, pos, nb = W.plan_send(lib, [500, 1, 1, 1, 1], [1, 1, 1, 1, 1], max_payload=100,
                                          batch_across_groups=False)
        assert nb == 2 and burst.tolist() == [0, 1, 1, 1, 1] and env.tolist() == [-1] * 5
        assert W.plan_send(lib, [], [])[3] == 0
    # the two implementations agree on random inputs
    rng = np.random.default_rng(3)
    for _ in range(50):
        n = int(rng.integers(0, 400))
        est = rng.integers(1, 3000, n)
        key = rng.integers(0, 6, n)
        mp = int(rng.integers(1, 20000))
        mb = int(rng.integers(0, 6))
        a = W.plan_send(libs[0], est, key, mp, mb, True)
        b = W.plan_send(libs[1], est, key, mp, mb, True)
        assert all(np.array_equal(x, y) for x, y in zip(a[:3], b[:3])) and a[3] == b[3]
