"""The accept-reply tail of PaxosCoordinatorState.main (PaxosCoordinatorState.java:1173-1213) with its
`Math.random() > 0.99` coin ENUMERATED: for every proposal still outstanding, every member answers
once, slot by slot, member by member, either in the coordinator's own ballot or - the coin - in a
higher one; main asserts after every reply that a non-null result has the right type and that the
decided / preempted slot has left myProposals, and at the end that myProposals is empty.

`model()` below follows the Java statement by statement (handleAcceptReplyMyBallot PCS:597-640,
recordSlotNumber :809-825, WaitforUtility :51-68, getMedianMinus :859-875, handleAcceptReplyHigherBallot
:661-675, and, because the engine is driven through gpx_accept_reply_batch = PISM.handleAcceptReply,
PaxosCoordinator.handleAcceptReply :210-250 with nullifyCoordinatorIfPreemptedFully PISM:1361-1364); it is
written here independently of oracle/gpx_oracle.cpp, so the test pins BOTH to the reference's text."""
import itertools

import numpy as np

from gigapaxos_amd import Engine, hri_create, S_OK, D_DECISION, D_PREEMPTED


def model(members, me, nprop, coins, maxcps=None, init_node_slots=None):
    """coins[(slot_index, member_index)] = True -> the reply carries ballot (myBallotNum + 1, me).
    maxcps[(slot_index, member_index)] = the reply's maxCheckpointedSlot (main passes -1 everywhere);
    init_node_slots = nodeSlotNumbers at the start (createHRI: zeros).
    Returns the list of (vote index, slot, bnum, bcoord, median, kind) outputs in arrival order."""
    K = len(members)
    my = (0, me)                       # createHRI: coordBallot (0, coordinator)
    node_slots = list(init_node_slots) if init_node_slots is not None else [0] * K  # createHRI: new int[members.length]
    proposals = {s: [False] * K for s in range(1, nprop + 1)}  # slot -> WaitforUtility.responded
    coordinator = True
    out = []
    v = 0
    for si, slot in enumerate(range(1, nprop + 1)):
        for j in range(K):
            ballot = (my[0] + 1, my[1]) if coins[(si, j)] else my
            maxcp = -1 if maxcps is None else maxcps[(si, j)]    # main passes -1
            if coordinator:
                if ballot > my:                                  # Ballot.compareTo > 0
                    if slot in proposals:                        # handleAcceptReplyHigherBallot
                        del proposals[slot]
                        out.append((v, slot, my[0], my[1], -1, D_PREEMPTED))
                    if not proposals:                            # isPreemptedFully -> coordinator = null
                        coordinator = False
                else:                                            # == my ballot
                    idx = -1
                    for q in range(K):                           # getIndex: last match
                        if members[q] == members[j]:
                            idx = q
                    if node_slots[idx] < maxcp:                  # recordSlotNumber, plain <
                        node_slots[idx] = maxcp
                    w = proposals.get(slot)
                    if w is not None:
                        w[idx] = True                            # updateHeardFrom
                        if sum(w) > K // 2:                      # heardFromMajority
                            srt = sorted(node_slots)             # getMedianMinus
                            med = srt[K // 2 - 1] if K % 2 == 0 else srt[K // 2]
                            del proposals[slot]
                            out.append((v, slot, my[0], my[1], med, D_DECISION))
            v += 1
    return out, proposals, coordinator, node_slots


def run_all(lib, K, nprop):
    """Every coin pattern for K members and nprop outstanding proposals through gpx_accept_reply_batch;
    returns the number of patterns checked."""
    members = list(range(21, 21 + 3 * K, 3))[:K]   # main: ascending ids starting at myID = 21
    me = members[0]
    keys = [(si, j) for si in range(nprop) for j in range(K)]
    checked = 0
    G = 1 << (len(keys))
    # one engine, one group per pattern: all patterns in ONE batch (votes of a group keep their order)
    e = Engine(lib, me, G, kmax=K, window=8, max_batch=G * len(keys) + 16)
    mem = np.tile(np.array(members, np.int32), (G, 1))
    assert (e.create_groups(np.arange(G), mem, K, hri_create(G, K, me)) == S_OK).all()
    for _ in range(nprop):
        sl, bn, bc, md, st = e.propose(np.arange(G, dtype=np.int32))
        assert (st == S_OK).all()
    cols = [[] for _ in range(6)]
    expect = []
    for p, bits in enumerate(itertools.product((False, True), repeat=len(keys))):
        coins = dict(zip(keys, bits))
        out, left, coord, _ = model(members, me, nprop, coins)
        # main's own assertions on the model's run
        assert not left or not coord                      # `assert (pcs.myProposals.isEmpty())`: every slot is
        # decided or preempted unless the coordinator resigned first (PISM level; main calls PCS directly)
        for (_, slot, _, _, _, kind) in out:
            assert kind in (D_DECISION, D_PREEMPTED)
        expect.append(out)
        for si, slot in enumerate(range(1, nprop + 1)):
            for j in range(K):
                hb = coins[(si, j)]
                for c, val in zip(cols, (p, 1 if hb else 0, me, slot, members[j], -1)):
                    c.append(val)
        checked += 1
    # interleave the groups' votes (stable per group) so that the batch is not sorted by group
    n = len(cols[0])
    order = np.argsort(np.arange(n) % len(keys), kind="stable")
    arrs = [np.array(c, np.int32)[order] for c in cols]
    d = e.accept_reply(*arrs)
    got = d.as_tuple_array()
    gi = 0
    for p, out in enumerate(expect):
        rows = got[gi:gi + len(out)]
        assert rows.shape[0] == len(out) and (rows[:, 0] == p).all(), f"pattern {p}: decision count"
        want = np.array([(p,) + o[1:] for o in out], np.int32).reshape(-1, 6)
        assert (rows == want).all(), f"pattern {p}: {rows.tolist()} != {want.tolist()}"
        gi += len(out)
    assert gi == got.shape[0]
    e.close()
    return checked


MAXCP_CHOICES = ("-1", "0", "slot-1", "slot")


def run_maxcp(lib, K, nprop, init_node_slots, sample=None, seed=0):
    """The same replay with recordSlotNumber (PCS:809-825) and a non-trivial getMedianMinus (PCS:859-875) in
    play: every vote also draws its maxCheckpointedSlot from {-1, 0, slot - 1, slot} and nodeSlotNumbers
    starts from `init_node_slots` (a hot-restored coordinator, PaxosCoordinator.java:122-131) - 8^(K * nprop)
    patterns, all of them or a seeded `sample`.  Checks the decided stream AND the final nodeSlotNumbers /
    coordinator flag of every group (late votes change the median of later decisions: PCS:607 runs before
    the pstate == null test of :618).  Returns the number of patterns checked."""
    members = list(range(21, 21 + 3 * K, 3))[:K]
    me = members[0]
    keys = [(si, j) for si in range(nprop) for j in range(K)]
    nk = len(keys)
    total = 8 ** nk
    if sample is not None and sample < total:
        picks = np.random.default_rng(seed).choice(total, size=sample, replace=False)
    else:
        picks = np.arange(total)
    G = int(picks.shape[0])
    e = Engine(lib, me, G, kmax=K, window=8, max_batch=G * nk + 16)
    mem = np.tile(np.array(members, np.int32), (G, 1))
    rows = hri_create(G, K, me)
    rows["node_slots"][:, :K] = np.array(init_node_slots, np.int32)
    assert (e.create_groups(np.arange(G), mem, K, rows) == S_OK).all()
    for _ in range(nprop):
        assert (e.propose(np.arange(G, dtype=np.int32))[4] == S_OK).all()
    cols = [np.zeros(G * nk, np.int32) for _ in range(6)]
    expect, final = [], []
    for p, code in enumerate(picks.tolist()):
        coins, maxcps = {}, {}
        for key in keys:                                   # one octal digit per vote: coin * 4 + maxcp choice
            d = code & 7
            code >>= 3
            coins[key] = bool(d >> 2)
            slot = key[0] + 1
            maxcps[key] = (-1, 0, slot - 1, slot)[d & 3]
        out, left, coord, ns = model(members, me, nprop, coins, maxcps, init_node_slots)
        expect.append(out)
        final.append((coord, ns))
        for v, key in enumerate(keys):
            i = p * nk + v
            cols[0][i] = p
            cols[1][i] = 1 if coins[key] else 0
            cols[2][i] = me
            cols[3][i] = key[0] + 1
            cols[4][i] = members[key[1]]
            cols[5][i] = maxcps[key]
    order = np.argsort(np.arange(G * nk) % nk, kind="stable")  # interleave the groups, stable per group
    d = e.accept_reply(*[c[order] for c in cols])
    got = d.as_tuple_array()
    gi = 0
    for p, out in enumerate(expect):
        rows_p = got[gi:gi + len(out)]
        want = np.array([(p,) + o[1:] for o in out], np.int32).reshape(-1, 6)
        assert rows_p.shape == want.shape and (rows_p == want).all(), \
            f"pattern {int(picks[p])}: {rows_p.tolist()} != {want.tolist()}"
        gi += len(out)
    assert gi == got.shape[0]
    snap, st = e.snapshot(np.arange(G))
    assert (st == S_OK).all()
    want_coord = np.array([1 if c else 0 for c, _ in final], np.int32)
    assert (snap["has_coord"] == want_coord).all()
    want_ns = np.array([ns for _, ns in final], np.int32)
    live = want_coord == 1                                  # HotRestoreInfo shows nodeSlots of a live coordinator only
    assert (snap["node_slots"][:, :K][live] == want_ns[live]).all()
    e.close()
    return G


def model_stream(members, me, nprop, votes, init_node_slots=None, base=0):
    """The same reading driven by an ARBITRARY vote stream instead of main's slot-by-slot, member-by-member loop:
    votes = [(slot, member index, ballot kind, maxCheckpointedSlot)], ballot kind -1 / 0 / +1 = lower than /
    equal to / higher than the coordinator's ballot (PaxosCoordinator.handleAcceptReply :210-250: a lower ballot
    is only logged), any slot (never proposed, decided or preempted long ago), duplicates, any order.
    Returns (outputs, outstanding slots, coordinator alive, nodeSlotNumbers)."""
    K = len(members)
    my = (0, me)
    node_slots = list(init_node_slots) if init_node_slots is not None else [0] * K
    proposals = {s: [False] * K for s in range(1, nprop + 1)}
    if base:                                                     # a coordinator restored at nextProposalSlot 1 + base (Java ints wrap)
        from tests.acc_enum_common import I32
        proposals = {int(I32(s) + base): w for s, w in proposals.items()}
    coordinator = True
    out = []
    for v, (slot, j, bkind, maxcp) in enumerate(votes):
        if not coordinator:                                      # PaxosCoordinator.handleAcceptReply(c == null)
            continue
        if bkind > 0:                                            # ballot.compareTo(getBallot()) > 0
            if slot in proposals:                                # handleAcceptReplyHigherBallot: myProposals.remove
                del proposals[slot]
                out.append((v, slot, my[0], my[1], -1, D_PREEMPTED))
            if not proposals:                                    # PISM.nullifyCoordinatorIfPreemptedFully (:1361-1364)
                coordinator = False
        elif bkind == 0:                                         # handleAcceptReplyMyBallot
            acceptor = members[j] if j >= 0 else -7              # j < 0: a node that is no member of the group
            for i in range(K):                                   # recordSlotNumber: every i with members[i] == acceptor
                if members[i] == acceptor and node_slots[i] < maxcp:
                    node_slots[i] = maxcp
            w = proposals.get(slot)
            if w is not None:
                idx = -1
                for q in range(K):                               # WaitforUtility.getIndex: last match
                    if members[q] == acceptor:
                        idx = q
                if 0 <= idx < K:                                 # updateHeardFrom (WaitforUtility.java:51-62)
                    w[idx] = True
                if sum(w) > K // 2:                              # heardFromMajority
                    srt = sorted(node_slots)                     # getMedianMinus
                    med = srt[K // 2 - 1] if K % 2 == 0 else srt[K // 2]
                    del proposals[slot]
                    out.append((v, slot, my[0], my[1], med, D_DECISION))
        # else: a reply to a lower ballot: nothing happens
    return out, proposals, coordinator, node_slots


def run_streams(lib, K, nprop, n_groups, n_votes, seed=0, p_higher=0.03, p_lower=0.08, p_stranger=0.0, p_extreme=0.0, base=0):
    """n_groups coordinators with nprop outstanding proposals each, every one fed its own random stream of
    n_votes accept replies (any member, any slot in [0, nprop + 1], lower / own / higher ballots, checkpoint
    slots -1 .. nprop) - all in ONE gpx_accept_reply_batch call, the groups interleaved; the decided stream,
    the coordinator's survival and nodeSlotNumbers of every group against model_stream."""
    rng = np.random.default_rng(seed)
    members = list(range(21, 21 + 3 * K, 3))[:K]
    me = members[0]
    init = [int(x) for x in rng.integers(0, 2, K)]
    G = n_groups
    e = Engine(lib, me, G, kmax=K, window=8, max_batch=G * n_votes + 16)
    mem = np.tile(np.array(members, np.int32), (G, 1))
    rows = hri_create(G, K, me)
    rows["node_slots"][:, :K] = np.array(init, np.int32)
    if base:     # the instance and its coordinator restored at slot 1 + base: the proposals' slots cross the int wrap
        from tests.acc_enum_common import I32
        rows["acc_slot"] = rows["next_proposal_slot"] = int(I32(1) + base)
        rows["acc_gc_slot"] = int(I32(-1) + base)
    assert (e.create_groups(np.arange(G), mem, K, rows) == S_OK).all()
    for _ in range(nprop):
        assert (e.propose(np.arange(G, dtype=np.int32))[4] == S_OK).all()
    n = G * n_votes
    slot = rng.integers(0, nprop + 2, n).astype(np.int32)
    mj = rng.integers(0, K, n).astype(np.int32)
    if p_stranger > 0.0:                                         # votes of a node that is no member (index -1)
        mj[rng.random(n) < p_stranger] = -1
    u = rng.random(n)
    bkind = np.where(u < p_higher, 1, np.where(u < p_higher + p_lower, -1, 0)).astype(np.int32)
    maxcp = rng.integers(-1, nprop + 1, n).astype(np.int32)
    if p_extreme > 0.0:      # checkpoint slots half the int range apart: recordSlotNumber compares with a PLAIN < (PCS:809-825)
        pick = rng.random(n) < p_extreme
        maxcp[pick] = rng.choice(np.array([-2**31, -2**31 + 1, -2**30, 2**30, 2**31 - 2, 2**31 - 1], np.int64), size=int(pick.sum())).astype(np.int32)
    if base:
        slot = (slot.astype(np.int64) + base + 2**31) % 2**32 - 2**31
        slot = slot.astype(np.int32)
    gcol = np.repeat(np.arange(G, dtype=np.int32), n_votes)
    bnum = np.where(bkind > 0, 1, 0).astype(np.int32)
    bcoord = np.where(bkind < 0, me - 1, me).astype(np.int32)
    acc = np.where(mj >= 0, np.array(members, np.int32)[np.maximum(mj, 0)], -7).astype(np.int32)
    expect, final = [], []
    for p in range(G):
        lo = p * n_votes
        votes = list(zip(slot[lo:lo + n_votes].tolist(), mj[lo:lo + n_votes].tolist(), bkind[lo:lo + n_votes].tolist(),
                         maxcp[lo:lo + n_votes].tolist()))
        out, _, coord, ns = model_stream(members, me, nprop, votes, init, base)
        expect.append(out)
        final.append((coord, ns))
    order = np.argsort(np.arange(n) % n_votes, kind="stable")    # interleave the groups, stable per group
    d = e.accept_reply(gcol[order], bnum[order], bcoord[order], slot[order], acc[order], maxcp[order])
    got = d.as_tuple_array()
    gi = 0
    for p, out in enumerate(expect):
        rows_p = got[gi:gi + len(out)]
        want = np.array([(p,) + o[1:] for o in out], np.int32).reshape(-1, 6)
        assert rows_p.shape == want.shape and (rows_p == want).all(), f"group {p}: {rows_p.tolist()} != {want.tolist()}"
        gi += len(out)
    assert gi == got.shape[0]
    snap, st = e.snapshot(np.arange(G))
    assert (st == S_OK).all()
    want_coord = np.array([1 if c else 0 for c, _ in final], np.int32)
    assert (snap["has_coord"] == want_coord).all()
    want_ns = np.array([ns for _, ns in final], np.int32)
    live = want_coord == 1
    assert (snap["node_slots"][:, :K][live] == want_ns[live]).all()
    e.close()
    return G
