"""The WHOLE round - propose -> ACCEPT at every replica -> accept replies -> decision -> BATCHED_COMMIT ->
in-order execution - against the two independent Python readings of the Java put together:

  coordinator: PCS.propose + initCommander (PaxosCoordinatorState.java:233-263, 841-851: the ACCEPT carries
               getMajorityCommittedSlot() = getMedianMinus(nodeSlotNumbers) :859-875) and the accept-reply side of
               tests/pcs_enum_common.model_stream (PaxosCoordinator.java:210-250, PCS:597-683, 809-825)
  acceptors:   tests/acc_enum_common.Acceptor (PaxosAcceptor.java:302-385, 462-506; PISM:1080-1166, 1432-1528,
               1619-1701); an ACCEPT_REPLY's maxCheckpointedSlot is what that replica reports (PISM:1137-1140)

  gaps:        getMaxCommittedSlot / getMissingCommittedSlots (PaxosAcceptor.java:405-438) and PISM.shouldSync
               (PISM:2341-2364) of every replica at the end, against gpx_gap_scan
  prepares:    PISM.handlePrepare -> PaxosAcceptor.handlePrepare (PaxosAcceptor.java:239-293) of every replica at
               the end (ballots below / at / above the acceptor's), against gpx_prepare_batch
  failover:    (failover=True) node 0 gone, replica 1 runs for coordinator of every group: makeCoordinator, the
               PREPAREs at the survivors, PISM.handlePrepareReply -> PaxosCoordinator.handlePrepareReply ->
               isPrepareAcceptedByMajority / combinePValuesOntoProposals / spawnCommandersForProposals (PCS:271-587;
               Candidate below), the view change's ACCEPTs at the survivors
  pauses:      (p_pause > 0) PISM.tryPause (PISM:2004-2035) of random instances between rounds: refused unless
               PaxosAcceptor.caughtUp (PaxosAcceptor.java:451-459) and PaxosCoordinator.caughtUp (PaxosCoordinator.java:
               369-371, PCS:758-761); the HotRestoreInfo of the others, and the instance hotRestore (PISM:677-690)
               makes of it again, which then plays on
  pokes:       (pokes=True) PISM.pokeLocalCoordinator (PISM:2268-2279) -> reissueAcceptIfWaitingTooLong
               (PaxosCoordinator.java:334-350) -> isCommandering / reInitCommander / initCommander (PCS:741-750, 841-851)
               minus the clock, after every round, against gpx_poke_scan: the one ACCEPT an active coordinator sends
               again is the one for its acceptor's next slot, with the median as it is NOW
  requests:    PISM.handleProposal's choice (PISM:817-888): propose iff PaxosCoordinator.exists(coordinator,
               paxosState.getBallot()) (PaxosCoordinator.java:168-174), else forward to getBallotCoord()

               and the stopped test of PISM.handlePaxosMessage (:456-460) in front of every message

Neither is written from oracle/gpx_oracle.cpp.  Three replicas per group, replica 0 the coordinator of every
group; per round every group proposes once or twice, every message (ACCEPT, reply, commit) of every replica is
lost with some probability, lost ACCEPTs and commits are sent again a round later (an ACCEPT that arrives after
its commit's placeholder: reconstructDecision; commits out of order: runs of several slots), replies reach the
coordinator in random order.  The libraries
(three engines behind the C-ABI: the oracle on the CPU, the HIP engine on the GPU) get the same traffic as
batches; every propose result, reply word, status, decision, execution run and final HotRestoreInfo row must
equal the model's."""
import numpy as np

from gigapaxos_amd import (Engine, hri_create, S_OK, S_FORWARD, S_REFUSED, S_STOPPED, S_BUSY, S_WINDOW, A_STOP, C_HASVALUE, C_STOP,
                           D_DECISION, D_PREEMPTED, RETIRE_PAUSE)

P_NACK, P_TOLOG = 1, 2   # GPX_P_NACK, GPX_P_TOLOG (include/gpx.h)
PV_STOP = 1              # GPX_PV_STOP
from tests.acc_enum_common import Acceptor, PValue, I32

WINDOW = 16


class Coordinator:
    """PaxosCoordinatorState as created by createHRI: ballot (0, me), active, nextProposalSlot 1."""

    def __init__(self, me, K):
        self.K = K
        self.my = (0, me)
        self.next = 1
        self.node_slots = [0] * K
        self.proposals = {}           # slot -> WaitforUtility.responded
        self.stops = set()            # outstanding proposals that are stop requests
        self.alive = True

    def median(self):                 # getMedianMinus
        K = self.K
        srt = sorted(self.node_slots)
        return srt[K // 2 - 1] if K % 2 == 0 else srt[K // 2]

    def propose(self, stop=False):
        """PCS.propose (:233-263) by an active coordinator -> (slot, bnum, bcoord, median), or None: "no point
        enqueuing anything after stop" (:235-239: the previous proposal is still outstanding and is a stop)"""
        if (self.next - 1) in self.proposals and (self.next - 1) in self.stops:
            return None
        slot = self.next
        self.next += 1
        self.proposals[slot] = [False] * self.K
        if stop:
            self.stops.add(slot)
        return (slot, self.my[0], self.my[1], self.median())     # initCommander: AcceptPacket(.., median)

    def reply(self, slot, j, ballot, maxcp):
        """one ACCEPT_REPLY of member j -> None | (slot, bnum, bcoord, median, kind)"""
        if not self.alive:
            return None
        if ballot > self.my:
            out = None
            if slot in self.proposals:
                del self.proposals[slot]
                self.stops.discard(slot)
                out = (slot, self.my[0], self.my[1], -1, D_PREEMPTED)
            if not self.proposals:
                self.alive = False
            return out
        if ballot < self.my:
            return None
        if self.node_slots[j] < maxcp:                           # recordSlotNumber (plain <), before the pstate test
            self.node_slots[j] = maxcp
        w = self.proposals.get(slot)
        if w is None:
            return None
        w[j] = True
        if sum(w) > self.K // 2:
            del self.proposals[slot]
            self.stops.discard(slot)
            return (slot, self.my[0], self.my[1], self.median(), D_DECISION)
        return None


def max_committed_slot(a):
    """PaxosAcceptor.getMaxCommittedSlot (PaxosAcceptor.java:425-438): TreeMap.lastKey() is the largest key in SIGNED
    order; only when that is Integer.MAX_VALUE does the Java walk the keys with the wraparound compare"""
    if a.stopped or not a.committed:
        return a._slot - 1
    max_slot = max(a.committed)
    if max_slot == 2**31 - 1:
        max_slot = a._slot - 1
        for i in sorted(a.committed):
            if i - max_slot > 0:
                max_slot = i
    return max_slot


def missing_committed_slots(a, size_limit):
    """PaxosAcceptor.getMissingCommittedSlots (:405-423): None for a stopped instance"""
    if a.stopped:
        return None
    missing = []
    maxc = max_committed_slot(a)
    i = a._slot
    while i - maxc < 0 and i - (a._slot + size_limit) < 0:
        # no commit, or a meta-commit without its accept
        if i not in a.committed or (not a.committed[i].has_value and i not in a.accepted):
            missing.append(i)
        i += 1
    return missing


def should_sync(a, threshold, mode):
    """PISM.shouldSync(getMaxCommittedSlot(), threshold, syncMode) (PISM:2341-2364); mode 0 default, 1 SYNC_TO_PAUSE,
    2 FORCE_SYNC"""
    max_decision = max_committed_slot(a)
    expected = a._slot
    nontrivial = max_decision - expected >= threshold // 100
    small = threshold <= 1
    return (max_decision - expected >= threshold or (expected in (0, 1) and (nontrivial or small)) or
            (nontrivial and mode == 1) or mode == 2)


def check_gaps(eng, acc, G, what):
    """gpx_gap_scan of every replica against the three readings above; returns the number of groups with a gap"""
    from gigapaxos_amd import wire as W
    gaps = 0
    for a, e in enumerate(eng):
        we = W.WireEngine(e)
        for threshold, mode, limit in ((2, 0, 64), (1, 0, 64), (300, 1, 5), (4, 2, 64), (3, 0, 2)):
            first, maxc, missing, sync, st = W.gap_scan(we, np.arange(G, dtype=np.int32), threshold, mode, limit)
            for g in range(G):
                m = acc[a][g]
                want = missing_committed_slots(m, limit)
                if want is None:
                    assert int(st[g]) == S_STOPPED and int(missing[g]) == 0, f"{what} replica {a} group {g}: stopped"
                    continue
                mask = 0
                for s_ in want:
                    mask |= 1 << (s_ - m._slot)
                gaps += bool(want) and threshold == 2
                assert (int(st[g]), int(first[g]), int(maxc[g]), int(missing[g]), int(sync[g])) == \
                    (S_OK, m._slot, max_committed_slot(m), mask, int(should_sync(m, threshold, mode))), \
                    f"{what} replica {a} group {g}: gap scan ({threshold}, {mode}, {limit})"
    return gaps


def handle_prepare(a, ballot, first_undecided):
    """PISM.handlePrepare (PISM:900-1006) -> PaxosAcceptor.handlePrepare (PaxosAcceptor.java:239-273) with
    pruneAcceptedProposals (:283-293) and getMaxGCSlotFirstUndecidedSlot (:275-280):
    -> None (stopped) | (reply ballot, gc slot, nack, to_log, [(slot, accepted ballot, is stop)] ascending)"""
    if a.stopped:
        return None
    first_undecided = type(a._slot)(first_undecided)       # (a Java int where the instance's slots are: acc_enum_common.I32)
    prev = a.ballot
    if ballot > a.ballot:                                  # strictly greater: adopt
        a.ballot = ballot
    nack = a.ballot > ballot                               # "send pvalues only if not NACKing"
    pvalues = [] if nack else sorted((s, pv.ballot, pv.stop) for s, pv in a.accepted.items() if s - first_undecided >= 0)
    gc = first_undecided - 1 if a.acceptedGCSlot - (first_undecided - 1) < 0 else a.acceptedGCSlot
    return (a.ballot, gc, nack, prev < a.ballot, pvalues)  # LogMessagingTask iff my ballot got upgraded (:975-983)


def check_prepares(eng, acc, G, nodes, rng):
    """gpx_prepare_batch of every replica - two PREPAREs per group, ballots below / equal to / above the acceptor's,
    firstUndecidedSlot around its slot - against handle_prepare; returns the number of pvalues carried"""
    carried = 0
    for a, e in enumerate(eng):
        for _ in range(2):
            pick = rng.integers(0, 4, G)
            bnum = np.array([0, 0, 1, 2], np.int32)[pick]
            bcoord = np.array([nodes[0] - 1, nodes[0], nodes[1 % len(nodes)], nodes[-1]], np.int32)[pick]
            first = np.array([acc[a][g]._slot for g in range(G)], np.int32) + rng.integers(-2, 3, G).astype(np.int32)
            (rb, rc, rg, rf, st), rows = e.prepare(np.arange(G, dtype=np.int32), bnum, bcoord, first)
            want_rows = []
            for g in range(G):
                out = handle_prepare(acc[a][g], (int(bnum[g]), int(bcoord[g])), int(first[g]))
                if out is None:
                    assert int(st[g]) == S_STOPPED, f"replica {a} group {g}: PREPARE to a stopped instance"
                    continue
                ballot, gc, nack, to_log, pv = out
                assert (int(st[g]), int(rb[g]), int(rc[g]), int(rg[g]), int(rf[g])) == \
                    (S_OK, ballot[0], ballot[1], gc, (P_NACK if nack else 0) | (P_TOLOG if to_log else 0)), \
                    f"replica {a} group {g}: PREPARE ({bnum[g]}, {bcoord[g]}) first {first[g]}"
                want_rows += [(g, s_, b[0], b[1]) for s_, b, _ in pv]
            assert rows == want_rows, f"replica {a}: accepted pvalues of the prepare replies"
            carried += len(want_rows)
    return carried


def value_handle(slot, ballot):
    """the caller's 64-bit key of the request value of an accepted pvalue (any injective function will do)"""
    return (slot << 24) | (ballot[0] << 12) | (ballot[1] & 0xfff)


def wraparound_max(values):
    """`if (maxSlot == null) maxSlot = cur; if (cur - maxSlot > 0) maxSlot = cur;` over the values in their order"""
    best = None
    for v in values:
        if best is None:
            best = v
        if v - best > 0:
            best = v
    return best


class Candidate:
    """A coordinator being elected: PaxosCoordinator.makeCoordinator (PaxosCoordinator.java:66-89) ->
    new PaxosCoordinatorState(bnum, me, acceptor slot, members, null) (PCS:168-181: nodeSlotNumbers = -1) -> prepare();
    then PISM.handlePrepareReply (PISM:1008-1068) -> PaxosCoordinator.handlePrepareReply (:264-310) ->
    isPreemptable, canIgnorePrepareReply, isPrepareAcceptedByMajority, combinePValuesOntoProposals,
    reproposePreemptedProposals, spawnCommandersForProposals, setCoordinatorActive (PCS:271-587), with pre-active
    proposals (PCS.propose while not active, :233-263) - without stop requests (processStop has nothing to do
    then)."""

    def __init__(self, K, ballot, slot):
        self.K = K
        self.my = ballot
        self.next = slot
        self.node_slots = [-1] * K
        self.carry = {}                 # carryoverProposals: slot -> (ballot, handle)
        self.heard = [False] * K        # waitforMyBallot
        self.active = False
        self.exists = True
        self.proposals = {}             # myProposals: slot -> (kind, handle)

    def median(self):
        srt = sorted(self.node_slots)
        return srt[self.K // 2 - 1] if self.K % 2 == 0 else srt[self.K // 2]

    def propose(self, handle, stop=False, kind="preactive"):
        """PCS.propose while not active (:233-263): the proposal gets the next slot, no ACCEPT goes out - or
        nothing at all behind a stop (:235-239) -> slot | None"""
        if (self.next - 1) in self.proposals and self.proposals[self.next - 1][2]:
            return None
        slot = self.next
        self.next += 1
        self.proposals[slot] = (kind, handle, stop)
        return slot

    def prepare_reply(self, j, rballot, gc, pvalues):
        """-> ('ignored' | 'recorded' | 'preempted' | 'elected', median, [(slot, 'carry' | 'noop' | 'preactive', handle)])"""
        gc = type(self.next)(gc)                           # (a Java int where the instance's slots are)
        if not self.exists:
            return ("ignored", 0, [])
        if not self.active and rballot > self.my:          # getPreActivesIfPreempted: the election is lost; the
            self.exists = False                            # pre-actives go to the winner (resignAsCoordinator)
            return ("preempted", 0, [(s_,) + self.proposals[s_] for s_ in sorted(self.proposals)])
        if self.active or rballot < self.my or self.heard[j]:   # canIgnorePrepareReply (waitforMyBallot == null once active)
            return ("ignored", 0, [])
        min_slot = gc + 1                                  # PrepareReplyPacket.getMinSlot: firstSlot or a lower accepted slot
        for s_, _, _, _ in pvalues:
            if s_ - min_slot < 0:
                min_slot = s_
        if self.node_slots[j] - min_slot < 0:              # recordSlotNumber(preply)
            self.node_slots[j] = min_slot
        for s_, b, h, stop in pvalues:                     # the pvalue of the highest ballot per slot
            if s_ not in self.carry or b > self.carry[s_][0]:
                self.carry[s_] = (b, h, stop)
        self.heard[j] = True
        if sum(self.heard) <= self.K // 2:
            return ("recorded", 0, [])
        if self.carry:                                     # combinePValuesOntoProposals (nothing to do without carry-overs)
            max_carry = wraparound_max(sorted(self.carry))     # getMaxPValueSlot (PCS:903-914) over the TreeMap's keys
            max_min = wraparound_max(self.node_slots)          # getMaxMinCarryoverSlot (PCS:921-931)
            pre = self.proposals                           # preActives = this.myProposals
            self.proposals = {}
            carried = {h for _, h, _ in self.carry.values()}
            cur = type(self.next)(max_min) - 1
            while (cur + 1) - max_carry <= 0:              # for (curSlot = maxMin; curSlot - maxCarryoverSlot <= 0; curSlot++)
                cur = cur + 1
                if cur in self.carry:                      # received pvalues dominate pre-active proposals
                    self.proposals[cur] = ("carry", self.carry[cur][1], self.carry[cur][2])
                elif cur not in pre:                       # no-op if neither received nor pre-active
                    self.proposals[cur] = ("noop", 0, False)
                else:                                      # stick with the pre-active unless a carry-over IS that request
                    if pre[cur][1] not in carried:         # isDuplicate: RequestPacket.equals = same handle
                        self.proposals[cur] = pre[cur]
                    del pre[cur]
            self.next = max_carry + 1
            for slot in sorted(pre):                       # reproposePreemptedProposals: TreeMap order, this.propose(..)
                self.propose(pre[slot][1], pre[slot][2])   # (nothing behind a stop)
            # processStop (:470-535).  Its two conversions compare the ballots of a stop and of a later request; every
            # proposal was re-stamped with MY ballot when it entered myProposals (ProposalStateAtCoordinator,
            # PCS:153-157), so the ballots are equal and only `assert (false)` is reached - a no-op in production.
            # What acts is the end: a stop somewhere but not last -> one more stop request behind everything.
            if any(st for _, _, st in self.proposals.values()) and not self.proposals[self.next - 1][2]:
                self.propose(0, True, "newstop")
        self.active = True                                 # spawnCommandersForProposals + setCoordinatorActive
        return ("elected", self.median(), [(s_,) + self.proposals[s_] for s_ in sorted(self.proposals)])


def check_failover(eng, acc, G, nodes, rng, K, p_drop, p_stop=0.0, p_dup_reply=0.0):
    """Node 0 is gone.  Replica 1 runs for coordinator of every group it still serves: gpx_election_begin, the
    PREPAREs at the survivors (handle_prepare above), their replies at the candidate (Candidate above), the
    ACCEPTs of the view change at the survivors (Acceptor.handleAccept) - every output against the readings.
    Returns (groups elected, ACCEPTs of the view change, carried, no-ops)."""
    from tests.election_common import (EB_PREPARING, V_IGNORED, V_RECORDED, V_ELECTED, V_PREEMPTED, E_CARRY, E_NOOP,
                                       E_PREACTIVE, E_NEWSTOP, S_PREACTIVE)
    kmap = {"carry": E_CARRY, "noop": E_NOOP, "preactive": E_PREACTIVE, "newstop": E_NEWSTOP}
    vmap = {"ignored": V_IGNORED, "recorded": V_RECORDED, "elected": V_ELECTED, "preempted": V_PREEMPTED}
    gs = np.array([g for g in range(G) if not acc[1][g].stopped], np.int32)
    bnum = np.array([acc[1][g].ballot[0] + 1 for g in gs.tolist()], np.int32)
    assert (eng[1].election_begin(gs, bnum) == EB_PREPARING).all()
    cand = {g: Candidate(K, (int(b), nodes[1]), acc[1][g]._slot) for g, b in zip(gs.tolist(), bnum.tolist())}
    first = np.array([cand[g].next for g in gs.tolist()], np.int32)   # PreparePacket(ballot, paxosState.getSlot())
    survivors = list(range(1, K))
    # client requests reach the candidate before it is elected: pre-active proposals (PCS.propose while not active);
    # some of them ARE requests a survivor has already accepted from the dead coordinator (forwarded again by their
    # clients): combinePValuesOntoProposals must not propose those twice
    fresh = 1 << 40
    n_pre = n_dup = 0
    for rep_ in range(2):
        sel = np.nonzero(rng.random(gs.shape[0]) < 0.35)[0]
        # (the engine keeps a proposal list of at most WINDOW slots and refuses a view change that needs more,
        # GPX_S_WINDOW; the Java's maps are unbounded: pre-active proposals only where the list stays shorter)
        span = {g: max([s_ - cand[g].next + 1 for a in survivors for s_ in acc[a][g].accepted] + [0]) for g in gs[sel].tolist()}
        sel = np.array([i for i in sel.tolist() if span[int(gs[i])] + 4 <= WINDOW], np.int64)
        if sel.shape[0] == 0:
            continue
        hs = []
        for g in gs[sel].tolist():
            known = [value_handle(s_, pv.ballot) for a in survivors for s_, pv in acc[a][g].accepted.items()
                     if s_ - cand[g].next >= 0 and not acc[a][g].stopped]
            if known and rng.random() < 0.4:
                hs.append(known[int(rng.integers(0, len(known)))])
                n_dup += 1
            else:
                fresh += 1
                hs.append(fresh)
        stops = (rng.random(sel.shape[0]) < p_stop * 5).astype(np.uint8)
        sl, bn, bc, md, st = eng[1].propose(gs[sel], stops, handle=np.array(hs, np.int64))
        for i, g in enumerate(gs[sel].tolist()):
            want = cand[g].propose(hs[i], bool(stops[i]))
            if want is None:
                assert int(st[i]) == S_REFUSED, f"failover: pre-active proposal behind a stop, group {g}"
                continue
            assert (int(sl[i]), int(bn[i]), int(bc[i]), int(st[i])) == (want,) + cand[g].my + (S_PREACTIVE,), \
                f"failover: pre-active proposal of group {g}"
        n_pre += sel.shape[0]
    order = survivors[:]
    rng.shuffle(order)
    # in some groups the last replica is running too, one ballot number higher: its own acceptor has adopted that
    # ballot already, so it answers the candidate's PREPARE with a NACK, and the candidate concedes (unless the
    # others made it coordinator first: then the late NACK is ignored)
    lg = gs[rng.random(gs.shape[0]) < 0.12]
    lg = np.array([g for g in lg.tolist() if not acc[K - 1][g].stopped], np.int32)
    if lg.shape[0]:
        lb = np.array([cand[g].my[0] + 1 for g in lg.tolist()], np.int32)
        (rb, rc, rg, rf, st), _ = eng[K - 1].prepare(lg, lb, np.full(lg.shape[0], nodes[K - 1], np.int32),
                                                     np.array([acc[K - 1][g]._slot for g in lg.tolist()], np.int32))
        for i, g in enumerate(lg.tolist()):
            ballot = handle_prepare(acc[K - 1][g], (int(lb[i]), nodes[K - 1]), acc[K - 1][g]._slot)[0]
            assert (int(st[i]), int(rb[i]), int(rc[i])) == (S_OK,) + ballot
    elected = {}
    preempted = {}
    for a in order:                                        # the PREPARE at replica a, its reply at the candidate
        keep = rng.random(gs.shape[0]) >= p_drop
        sub, sb, sf = gs[keep], bnum[keep], first[keep]
        if sub.shape[0] == 0:
            continue
        (rb, rc, rg, rf, st), rows = eng[a].prepare(sub, sb, np.full(sub.shape[0], nodes[1], np.int32), sf)
        pvs = [[] for _ in range(sub.shape[0])]
        want_rows = []
        replies = []
        for i, g in enumerate(sub.tolist()):
            out = handle_prepare(acc[a][g], (int(sb[i]), nodes[1]), int(sf[i]))
            if out is None:
                assert int(st[i]) == S_STOPPED
                continue
            ballot, gc, nack, to_log, pv = out
            assert (int(st[i]), int(rb[i]), int(rc[i]), int(rg[i]), int(rf[i])) == \
                (S_OK, ballot[0], ballot[1], gc, (P_NACK if nack else 0) | (P_TOLOG if to_log else 0)), f"failover: PREPARE at replica {a} group {g}"
            want_rows += [(i, s_, b[0], b[1]) for s_, b, _ in pv]
            pvs[i] = [(s_, b[0], b[1], value_handle(s_, b), PV_STOP if stop else 0) for s_, b, stop in pv]
            replies.append(i)
        assert rows == want_rows, f"failover: pvalues of replica {a}'s prepare replies"
        idx = np.array(replies, np.int64)
        if idx.shape[0] == 0:
            continue
        (vk, em, rst), lists = eng[1].prepare_reply(sub[idx], np.full(idx.shape[0], nodes[a], np.int32), rb[idx], rc[idx],
                                                    rg[idx] + 1, [pvs[i] for i in replies])
        for q, i in enumerate(replies):
            g = int(sub[i])
            kind, med, lst = cand[g].prepare_reply(a, (int(rb[i]), int(rc[i])), int(rg[i]),
                                                   [(s_, (b0, b1), h, bool(fl)) for s_, b0, b1, h, fl in pvs[i]])
            assert int(rst[q]) == S_OK and int(vk[q]) == vmap[kind], \
                f"failover: reply of replica {a} for group {g}: {kind} {lst} - status {int(rst[q])} kind {int(vk[q])} {lists[q]}"
            if kind == "preempted":
                got = [(s_, k_, h, bool(fl & PV_STOP)) for s_, k_, h, fl in lists[q]]
                assert got == [(s_, kmap[k_], h, stop) for s_, k_, h, stop in lst], f"failover: pre-actives of the preempted group {g}"
                preempted[g] = len(lst)
            if kind == "elected":
                assert int(em[q]) == med, f"failover: median of group {g}"
                got = [(s_, k_, h if k_ in (E_CARRY, E_PREACTIVE) else 0, bool(fl & PV_STOP)) for s_, k_, h, fl in lists[q]]
                assert got == [(s_, kmap[k_], h, stop) for s_, k_, h, stop in lst], f"failover: ACCEPTs of group {g}: {got} != {lst}"
                elected[g] = (med, lst)
        if p_dup_reply > 0.0:
            # a retransmitted PREPARE_REPLY: canIgnorePrepareReply (PCS:285-316) - the acceptor has answered already,
            # or the election is over either way (waitforMyBallot == null / the coordinator is gone)
            dup = [i for i in replies if rng.random() < p_dup_reply]
            if dup:
                di = np.array(dup, np.int64)
                (vk, em, rst), lists = eng[1].prepare_reply(sub[di], np.full(di.shape[0], nodes[a], np.int32), rb[di], rc[di],
                                                            rg[di] + 1, [pvs[i] for i in dup])
                for q, i in enumerate(dup):
                    g = int(sub[i])
                    kind, _, _ = cand[g].prepare_reply(a, (int(rb[i]), int(rc[i])), int(rg[i]),
                                                       [(s_, (b0, b1), h, bool(fl)) for s_, b0, b1, h, fl in pvs[i]])
                    assert kind == "ignored" and int(rst[q]) == S_OK and int(vk[q]) == V_IGNORED and lists[q] == [], \
                        f"failover: repeated reply of replica {a} for group {g}"
                check_failover.dup_replies = getattr(check_failover, "dup_replies", 0) + len(dup)
    # the ACCEPTs of the view change at the survivors, in the new ballot
    n_acc = n_carry = n_noop = 0
    votes = []                                             # their replies, on the way to the new coordinator
    for a in survivors:
        recs = [(g, s_, cand[g].my[0], cand[g].my[1], med, int(stop)) for g, (med, lst) in elected.items() for s_, _, _, stop in lst]
        n_carry += sum(k_ == "carry" for _, (_, lst) in elected.items() for _, k_, _, _ in lst) if a == 1 else 0
        n_noop += sum(k_ != "carry" for _, (_, lst) in elected.items() for _, k_, _, _ in lst) if a == 1 else 0
        if not recs:
            continue
        cols = np.array(recs, np.int32)
        (rb, rc, rm, rf, st), runs = eng[a].accept(cols[:, 0], cols[:, 2], cols[:, 3], cols[:, 1], cols[:, 4],
                                                   (cols[:, 5] * A_STOP).astype(np.uint8))
        want_runs = []
        for i, (g, s_, b0, b1, med, stop) in enumerate(recs):
            s_, med = type(acc[a][g]._slot)(s_), type(acc[a][g]._slot)(med)
            status, wb, wc, wm, wf, run = acc[a][g].handleAccept(PValue((b0, b1), s_, med, True, bool(stop)))
            assert (int(st[i]), int(rb[i]), int(rc[i]), int(rm[i]), int(rf[i])) == (status, wb, wc, wm, wf), \
                f"failover: ACCEPT {recs[i]} at replica {a}"
            if run is not None:
                want_runs.append((g, i, run[0], run[1]))
            if status == S_OK:
                votes.append((g, s_, a, wb, wc, wm))
        want_runs.sort(key=lambda t: (t[0], t[1]))
        got = runs.as_tuple_array()
        exp = np.array([(g, f, c) for g, _, f, c in want_runs], np.int32).reshape(-1, 3)
        assert got.shape == exp.shape and (got == exp).all(), f"failover: execution runs at replica {a}"
        n_acc += len(recs)
    # the new coordinators' rows
    eg = np.array(sorted(elected), np.int32)
    if eg.shape[0]:
        snap, st = eng[1].snapshot(eg)
        assert (st == S_OK).all() and (snap["has_coord"] == 1).all()
        assert (snap["coord_bnum"] == np.array([cand[g].my[0] for g in eg.tolist()], np.int32)).all()
        assert (snap["coord_bcoord"] == nodes[1]).all()
        assert (snap["next_proposal_slot"] == np.array([cand[g].next for g in eg.tolist()], np.int32)).all()
        assert (snap["node_slots"][:, :K] == np.array([cand[g].node_slots for g in eg.tolist()], np.int32)).all()
    check_failover.preactive = (n_pre, n_dup, sum(k_ == "preactive" for _, (_, lst) in elected.items() for _, k_, _, _ in lst))
    check_failover.newstops = sum(k_ == "newstop" for _, (_, lst) in elected.items() for _, k_, _, _ in lst)
    check_failover.preempted = (len(preempted), sum(preempted.values()))
    check_failover.votes = votes
    check_failover.elected = {g: cand[g] for g in elected}
    pg = np.array(sorted(preempted), np.int32)
    if pg.shape[0]:                                        # PISM.handlePrepareReply: this.coordinator = null
        assert (eng[1].snapshot(pg)[0]["has_coord"] == 0).all()
    return len(elected), n_acc, n_carry, n_noop


def run_rounds(lib, G, rounds, seed, p_drop=0.12, p_double=0.3, K=3, p_rival=0.0, p_stop=0.0, from_disk=True, failover=False,
               rounds_after=0, p_pause=0.0, pokes=False, p_dup_reply=0.0, base=0):
    """K replicas per group (nodes 100 .. 100 + K - 1, node 100 the coordinator).  Returns (records compared,
    slots executed over all replicas)."""
    rng = np.random.default_rng(seed)
    NODES = list(range(100, 100 + K))
    # from_disk = PaxosAcceptor.GET_ACCEPTED_PVALUES_FROM_DISK (:75-76) = the engine's GPX_F_ACCEPTS_FROM_DISK: an executed
    # slot's accept leaves acceptedProposals at once (true) or only when garbage collection reaches it (false)
    eng = [Engine(lib, NODES[a], G, kmax=K, window=WINDOW, max_batch=32 * G + 64, flags=1 if from_disk else 0)
           for a in range(K)]
    mem = np.tile(np.array(NODES, np.int32), (G, 1))
    rows0 = hri_create(G, K, NODES[0])
    first_slot = 1
    if base:
        # every instance restored (HotRestoreInfo) at slot 1 + base, Java ints wrapping (acc_enum_common.I32): with base
        # just below 2^31 the slots, the checkpoint slots and the medians cross Integer.MAX_VALUE during the rounds
        # (recordSlotNumber's plain < freezes nodeSlotNumbers at the wrap - PCS:809-825, the Java's own behaviour - and with
        # them the medians and every acceptedGCSlot: a retransmitted ACCEPT of an executed slot is then never collected.
        # The Java's map is unbounded; the engine's ring is not, so here the traffic stops WINDOW slots after the last GC)
        # No view change here: the Java's own is not sane at the wrap.  nodeSlotNumbers starts at the sentinel -1
        # (PCS:168-181) and getMaxMinCarryoverSlot compares with it by subtraction: a reply whose minimum slot is
        # exactly Integer.MAX_VALUE loses against -1 (MAX_VALUE - (-1) overflows), and combinePValuesOntoProposals then
        # walks from -1 to the highest carried slot - two billion no-ops; past the wrap every (negative) slot loses
        # against the sentinel and the carried pvalues are dropped.  (The libraries refuse such a view change with
        # GPX_S_WINDOW, include/gpx.h; this reading would walk the two billion slots.)
        assert not failover, "the reference's view change assumes 0 <= slot < Integer.MAX_VALUE"
        first_slot = I32(1) + base
        rows0["acc_slot"] = rows0["next_proposal_slot"] = int(first_slot)
        rows0["acc_gc_slot"] = int(first_slot - 2)
    for e in eng:
        assert (e.create_groups(np.arange(G), mem, K, rows0) == S_OK).all()
    coord = [Coordinator(NODES[0], K) for _ in range(G)]
    acc = [[Acceptor(first_slot, (0, NODES[0]), first_slot - 2, from_disk=from_disk) for _ in range(G)] for _ in range(K)]
    for c in coord:
        c.next = first_slot
    J = type(first_slot)        # what a slot read back from a batch column becomes again: int, or the Java int (I32)
    pending = [[] for _ in range(K)]        # per replica: ACCEPTs lost on their way, to be sent again
    pending_c = [[] for _ in range(K)]      # ... and commits
    forwarded = refused = stopped_props = 0
    paused = paused_coord = busy = relogged = poked = 0
    stop_slots = set()                      # (group, slot) of the proposals that are STOP requests
    checked = 0

    def check_runs(runs, want, what):
        got = runs.as_tuple_array()
        exp = np.array(want, np.int32).reshape(-1, 3)
        assert got.shape == exp.shape and (got == exp).all(), f"{what}: execution runs\n{got[:8]}\n{exp[:8]}"

    def poke(r, ci, coord):
        nonlocal checked, poked
        gs = np.array([g for g in range(G) if coord[g] is not None], np.int32)
        pk, sl, bn, bc, md, fl, hd, st = eng[ci].poke_scan(gs)
        for i, g in enumerate(gs.tolist()):
            c, s_ = coord[g], acc[ci][g]._slot
            want = (0, 0, 0, 0, 0, 0, 0)                                  # GPX_POKE_NONE
            if c.alive and s_ in c.proposals:                            # isActive() && isCommandering(slot)
                want = (1, s_, c.my[0], c.my[1], c.median(), PV_STOP if s_ in c.stops else 0,
                        sum(1 << j for j, heard in enumerate(c.proposals[s_]) if heard))
                poked += 1
            assert (int(pk[i]), int(sl[i]), int(bn[i]), int(bc[i]), int(md[i]), int(fl[i]), int(hd[i])) == want and \
                int(st[i]) == S_OK, f"round {r}: poke of group {g}"
        checked += gs.shape[0]

    def pause(r, ci, replicas, coord):
        """PaxosManager's deactivation of idle instances: tryPause, and (here: at once) the restore from what it left"""
        nonlocal checked, paused, paused_coord, busy, relogged
        for a in replicas:
            gs = [g for g in np.nonzero(rng.random(G) < p_pause)[0].tolist()
                  if not acc[a][g].stopped and (a != ci or coord[g] is not None)]
            if not gs:
                continue
            rows, st = eng[a].retire_groups(np.array(gs, np.int32), RETIRE_PAUSE)
            back = []
            for i, g in enumerate(gs):
                m = acc[a][g]
                c = coord[g] if a == ci and coord[g].alive else None        # PaxosCoordinator.caughtUp: c == null || ...
                caught = not m.committed and (not m.accepted or from_disk) and (c is None or not c.proposals)
                if not caught:
                    busy += 1
                    assert int(st[i]) == S_BUSY, f"round {r} replica {a}: pause of group {g} that is not caught up"
                    continue
                assert int(st[i]) == S_OK, f"round {r} replica {a}: pause of group {g}"
                row = rows[i]
                assert (int(row["acc_slot"]), int(row["acc_bnum"]), int(row["acc_bcoord"]), int(row["acc_gc_slot"])) == \
                    tuple(m.row()), f"round {r} replica {a} group {g}: acceptor part of the HotRestoreInfo"
                if c is None:                                               # getBallotIfActive & co. (PISM:2015-2018)
                    assert (int(row["has_coord"]), int(row["next_proposal_slot"])) == (0, -1)
                else:
                    assert (int(row["has_coord"]), int(row["coord_bnum"]), int(row["coord_bcoord"]), int(row["next_proposal_slot"]),
                            row["node_slots"][:K].tolist()) == (1, c.my[0], c.my[1], c.next, c.node_slots), \
                        f"round {r} group {g}: coordinator part of the HotRestoreInfo"
                    paused_coord += 1
                back.append(i)
                # "pause/unpause will lose acceptedProposals state, which is okay iff we always return accepted pvalues
                # from disk" (PaxosAcceptor.java:453-457).  The disk is the host's; here the coordinator's retransmission
                # brings the ACCEPTs back (same ballot, same slot: accepted again, or refused if the ballot rose since)
                for s_, pv in sorted(m.accepted.items()):
                    pending[a].append((g, s_, pv.ballot[0], pv.ballot[1], pv.median, int(pv.stop)))
                    relogged += 1
                m.accepted = {}
            if back:
                bi = np.array(back)
                assert (eng[a].create_groups(np.array(gs, np.int32)[bi], mem[:len(back)], K, rows[bi]) == S_OK).all()
                paused += len(back)
            checked += len(gs)

    def play(rounds_, ci, replicas, coord, p_rival_, first_votes, tag):
        """rounds_ rounds with replica ci as the coordinator of every group that has one in `coord` (None: no
        proposals there), messages only among `replicas`; first_votes = accept replies already on their way"""
        nonlocal checked, forwarded, refused, stopped_props, paused, busy, relogged
        for r_ in range(rounds_):
            r = f"{tag}{r_}"
            accepts = []                        # (g, slot, bnum, bcoord, median) of this round, in proposal order
            for rep in range(2):
                gs = np.arange(G, dtype=np.int32) if rep == 0 else np.nonzero(rng.random(G) < p_double)[0].astype(np.int32)
                # keep the coordinator's window: at most WINDOW - 2 outstanding proposals per group
                # ... and the acceptors' (the engine's rings hold WINDOW slots from the slowest replica's next slot on; the
                # Java's maps are unbounded: the model has no such limit, so the traffic stays inside it)
                # (with accepts kept in memory - from_disk false - an executed slot's accept holds its ring entry until
                # garbage collection reaches it: the oldest live slot of a replica is then acceptedGCSlot + 1)
                # (the oldest outstanding proposal too: its ring entry is the one slot next - WINDOW would need)
                # (written as distances from the next proposal slot, a - b, so that they hold at the int wrap as well)
                gs = np.array([g for g in gs.tolist() if coord[g] is not None and len(coord[g].proposals) < WINDOW - 2 and
                               all(coord[g].next - s_ < WINDOW - 1 for s_ in coord[g].proposals) and
                               all(coord[g].next - acc[a][g]._slot < WINDOW - 2 and
                                   ((from_disk and not base) or coord[g].next - (acc[a][g].acceptedGCSlot + 1) < WINDOW - 2) for a in replicas)],
                              np.int32)
                if gs.shape[0] == 0:
                    continue
                stop_req = (rng.random(gs.shape[0]) < p_stop).astype(np.uint8)
                sl, bn, bc, md, st = eng[ci].propose(gs, stop_req)
                for i, g in enumerate(gs.tolist()):
                    if acc[ci][g].stopped:                              # PISM.handlePaxosMessage :456-460
                        stopped_props += 1
                        assert int(st[i]) == S_STOPPED, f"round {r}: proposal to a stopped instance {g}"
                        continue
                    # PISM.handleProposal (:817-888): propose iff PaxosCoordinator.exists(coordinator, paxosState.getBallot())
                    # (PaxosCoordinator.java:168-174: there is one and its ballot is not below the local acceptor's),
                    # else the request is unicast to paxosState.getBallotCoord()
                    if coord[g].alive and coord[g].my >= acc[ci][g].ballot:
                        want = coord[g].propose(bool(stop_req[i]))
                        if want is None:
                            refused += 1
                            assert int(st[i]) == S_REFUSED, f"round {r}: proposal after a stop {g}"
                            continue
                        assert (int(sl[i]), int(bn[i]), int(bc[i]), int(md[i]), int(st[i])) == want + (S_OK,), \
                            f"round {r}: propose {g}: {(int(sl[i]), int(bn[i]), int(bc[i]), int(md[i]), int(st[i]))} != {want}"
                        accepts.append((g,) + want + (int(stop_req[i]),))
                        if stop_req[i]:
                            stop_slots.add((g, want[0]))
                    else:
                        forwarded += 1
                        assert (int(bn[i]), int(bc[i]), int(st[i])) == acc[ci][g].ballot + (S_FORWARD,), f"round {r}: forward {g}"
                checked += gs.shape[0]
            votes = first_votes if r_ == 0 else []   # (g, slot, member, bnum, bcoord, maxcp)
            rival = []
            if p_rival_ > 0.0:
                # a rival (node 101, ballot (1, 101)) pushes an ACCEPT of its own for the group's newest slot at the
                # replicas it reaches: their ballots rise, the coordinator's later ACCEPTs there are answered with the
                # higher ballot (PaxosAcceptor.acceptAndUpdateBallot :302-322), which preempts its proposals and, once
                # none is left, makes it resign (PCS:661-683, PISM:1361-1364)
                for g in np.nonzero(rng.random(G) < p_rival_)[0].tolist():
                    if coord[g] is not None and coord[g].next - first_slot > 0:
                        rival.append((g, coord[g].next - 1, 1, NODES[1], first_slot - 2, 0))
            for a in replicas:
                # (at replica 0 as well: once the coordinator's OWN acceptor has adopted the rival's ballot, requests
                # are forwarded to the rival instead of being proposed)
                mine = [t for t in rival if rng.random() < (0.25 if a == 0 else 0.6) and t[1] - acc[a][t[0]]._slot >= 0]
                todo = pending[a] + accepts + mine  # the retransmissions first, then this round's, then the rival's
                pending[a] = []
                lost = rng.random(len(todo)) < p_drop
                send = [t for t, l in zip(todo, lost) if not l]
                pending[a] = [t for t, l in zip(todo, lost) if l]
                if not send:
                    continue
                # a group's ACCEPTs keep their slot order, the groups are shuffled among each other
                by_group = {}
                for t in send:
                    by_group.setdefault(t[0], []).append(t)
                seq = []
                keys = list(by_group)
                rng.shuffle(keys)
                cursors = {g: 0 for g in keys}
                live = keys[:]
                while live:                      # round robin over the shuffled groups: interleaved, per-group order kept
                    nxt = []
                    for g in live:
                        seq.append(by_group[g][cursors[g]])
                        cursors[g] += 1
                        if cursors[g] < len(by_group[g]):
                            nxt.append(g)
                    live = nxt
                cols = np.array(seq, np.int32)
                (rb, rc, rm, rf, st), runs = eng[a].accept(cols[:, 0], cols[:, 2], cols[:, 3], cols[:, 1], cols[:, 4],
                                                           (cols[:, 5] * A_STOP).astype(np.uint8))
                want_runs = []
                for i, (g, slot, bnum, bcoord, median, stop) in enumerate(seq):
                    slot, median = J(slot), J(median)
                    m_ = acc[a][g]
                    if (not m_.stopped and (bnum, bcoord) >= m_.ballot and slot - m_.acceptedGCSlot > 0 and
                            any(k_ != slot and ((int(k_) ^ int(slot)) & (WINDOW - 1)) == 0 for k_ in m_.accepted)):
                        # the engine's limit, not the Java's (include/gpx.h): accepted pvalues live in a ring of `window`
                        # entries; an ACCEPT that would be stored where another live slot sits is dropped whole, like a
                        # lost packet - the coordinator's comes again, the rival's does not
                        assert int(st[i]) == S_WINDOW, f"round {r} replica {a}: ACCEPT {seq[i]} into an occupied ring entry"
                        if coord[g] is not None and (bnum, bcoord) == coord[g].my:
                            pending[a].append(seq[i])
                        continue
                    status, wb, wc, wm, wf, run = acc[a][g].handleAccept(PValue((bnum, bcoord), slot, median, True, bool(stop)))
                    assert (int(st[i]), int(rb[i]), int(rc[i]), int(rm[i]), int(rf[i])) == (status, wb, wc, wm, wf), \
                        f"round {r} replica {a}: ACCEPT {seq[i]}: got {(int(st[i]), int(rb[i]), int(rc[i]), int(rm[i]), int(rf[i]))}, the reading gives {(status, wb, wc, wm, wf)}"
                    if run is not None:
                        want_runs.append((g, i, run[0], run[1]))
                    if status == S_OK and coord[g] is not None and (bnum, bcoord) == coord[g].my:   # (the rival's replies go to the rival)
                        votes.append((g, slot, a, wb, wc, wm))
                want_runs.sort(key=lambda t: (t[0], t[1]))
                check_runs(runs, [(g, f, c) for g, _, f, c in want_runs], f"round {r} replica {a} accept")
                checked += len(seq)
            # the replies reach the coordinator in random order, some never
            votes = [v for v in votes if rng.random() >= p_drop]
            perm = rng.permutation(len(votes))
            votes = [votes[i] for i in perm]
            decisions = []
            if votes:
                cols = np.array(votes, np.int32)
                d = eng[ci].accept_reply(cols[:, 0], cols[:, 3], cols[:, 4], cols[:, 1], np.array(NODES, np.int32)[cols[:, 2]], cols[:, 5])
                want = []
                for i, (g, slot, a, wb, wc, wm) in enumerate(votes):
                    if acc[ci][g].stopped:                              # PISM.handlePaxosMessage :456-460: dropped
                        assert int(d.status[i]) == S_STOPPED, f"round {r}: vote for a stopped instance {g}"
                        continue
                    assert int(d.status[i]) == S_OK
                    out = coord[g].reply(slot, a, (wb, wc), wm)
                    if out is not None:
                        want.append((g, i) + out)
                want.sort(key=lambda t: (t[0], t[1]))
                exp = np.array([(t[0],) + t[2:] for t in want], np.int32).reshape(-1, 6)
                got = d.as_tuple_array()
                assert got.shape == exp.shape and (got == exp).all(), f"round {r}: decisions"
                decisions = [t for t in want if t[6] == D_DECISION]
                checked += len(votes)
            # BATCHED_COMMITs to every replica, some lost; a lost one comes again a round later as a full DECISION (its
            # request value with it: what a replica gets back when it asks for missing decisions, PISM:1432-1478)
            for a in replicas:
                todo = pending_c[a] + [(t[0], t[2], t[3], t[4], t[5], 0) for t in decisions]  # g, slot, bnum, bcoord, median, kind
                lost = rng.random(len(todo)) < p_drop
                send = [t for t, l in zip(todo, lost) if not l]
                pending_c[a] = [t[:5] + (C_HASVALUE | (C_STOP if (t[0], t[1]) in stop_slots else 0),)
                                for t, l in zip(todo, lost) if l]
                if not send:
                    continue
                cols = np.array(send, np.int32)
                st, runs = eng[a].commit(cols[:, 0], cols[:, 2], cols[:, 3], cols[:, 1], cols[:, 4], cols[:, 5].astype(np.uint8))
                want_runs = []
                for i, (g, slot, bnum, bcoord, median, kind) in enumerate(cols.tolist()):
                    slot, median = J(slot), J(median)
                    if not acc[a][g].stopped and slot - acc[a][g]._slot >= WINDOW:
                        # the engine's limit, not the Java's (include/gpx.h): decisions are kept for `window` slots from the
                        # next one to execute; a commit further ahead is dropped like a lost packet - and comes again
                        assert int(st[i]) == S_WINDOW, f"round {r} replica {a}: commit {cols[i]} beyond the window"
                        pending_c[a].append((g, slot, bnum, bcoord, median, C_HASVALUE | (C_STOP if (g, slot) in stop_slots else 0)))
                        continue
                    if kind & C_HASVALUE:
                        status, run = acc[a][g].handleDecision((bnum, bcoord), slot, median, bool(kind & C_STOP))
                    else:
                        status, run = acc[a][g].handleBatchedCommitSlot((bnum, bcoord), slot, median)
                    assert int(st[i]) == status, f"round {r} replica {a}: commit {cols[i]}"
                    if run is not None:
                        want_runs.append((g, i, run[0], run[1]))
                want_runs.sort(key=lambda t: (t[0], t[1]))
                check_runs(runs, [(g, f, c) for g, _, f, c in want_runs], f"round {r} replica {a} commit")
                checked += len(send)
            if pokes:
                poke(r, ci, coord)
            if p_pause > 0.0:
                pause(r, ci, replicas, coord)

    play(rounds, 0, list(range(K)), coord, p_rival, [], "round ")
    run_rounds.gaps = check_gaps(eng, acc, G, "final")
    run_rounds.failover = check_failover(eng, acc, G, NODES, rng, K, p_drop, p_stop, p_dup_reply) if failover else None
    coord2 = None
    if failover and rounds_after:
        # the new coordinators (replica 1) go on: the replies to the view change's ACCEPTs, decisions, commits,
        # executions, new proposals - among the survivors
        coord2 = [None] * G
        for g, cd in check_failover.elected.items():
            c2 = Coordinator(NODES[1], K)
            c2.my, c2.next, c2.node_slots = cd.my, cd.next, list(cd.node_slots)
            c2.proposals = {s_: [False] * K for s_ in cd.proposals}
            c2.stops = {s_ for s_, (_, _, stop) in cd.proposals.items() if stop}
            stop_slots.update((g, s_) for s_ in c2.stops)
            coord2[g] = c2
        before = checked
        play(rounds_after, 1, list(range(1, K)), coord2, 0.0, check_failover.votes, "after the view change ")
        run_rounds.after = checked - before
    run_rounds.carried = check_prepares(eng, acc, G, NODES, rng)   # (raises acceptor ballots: the final rows below see it)
    # final rows: acceptor side of every replica, coordinator side of replica 0
    for a in range(K):
        snap, st = eng[a].snapshot(np.arange(G))
        assert (st == S_OK).all()
        want = np.array([acc[a][g].row() for g in range(G)], np.int32)
        got = np.stack([snap["acc_slot"], snap["acc_bnum"], snap["acc_bcoord"], snap["acc_gc_slot"]], axis=1)
        assert (got == want).all(), f"replica {a}: acceptor rows"
    snap, _ = eng[0].snapshot(np.arange(G))
    alive = np.array([c.alive for c in coord])
    assert ((snap["has_coord"] != 0) == alive).all(), "coordinators that resigned"
    assert (snap["next_proposal_slot"][alive] == np.array([c.next for c in coord], np.int32)[alive]).all()
    assert (snap["node_slots"][:, :K][alive] == np.array([c.node_slots for c in coord], np.int32)[alive]).all()
    if coord2 is not None:
        eg = np.array([g for g in range(G) if coord2[g] is not None], np.int32)
        if eg.shape[0]:
            snap1, _ = eng[1].snapshot(eg)
            alive2 = np.array([coord2[g].alive for g in eg.tolist()])
            assert ((snap1["has_coord"] != 0) == alive2).all(), "new coordinators that resigned"
            assert (snap1["next_proposal_slot"][alive2] == np.array([coord2[g].next for g in eg.tolist()], np.int32)[alive2]).all()
            assert (snap1["node_slots"][:, :K][alive2] == np.array([coord2[g].node_slots for g in eg.tolist()], np.int32)[alive2]).all()
    executed = sum(int(acc[a][g]._slot - first_slot) for a in range(K) for g in range(G))
    for e in eng:
        e.close()
    run_rounds.resigned = int((~alive).sum())
    run_rounds.forwarded = forwarded
    run_rounds.refused = refused
    run_rounds.stopped_props = stopped_props
    run_rounds.stopped = sum(acc[a][g].stopped for a in range(K) for g in range(G))
    run_rounds.paused, run_rounds.paused_coord, run_rounds.busy, run_rounds.relogged = paused, paused_coord, busy, relogged
    run_rounds.poked = poked
    return checked, executed
