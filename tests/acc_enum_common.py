"""The ACCEPTOR side read from the Java, statement by statement, and checked by bounded exhaustive
enumeration.  Written from the reference's text, NOT from oracle/gpx_oracle.cpp, so that the test pins
both the oracle and the engine to it:

  PaxosInstanceStateMachine.handlePaxosMessage's stopped test        PISM:456-460
  PISM.handleAccept                                                  PISM:1080-1166
  PaxosAcceptor.acceptAndUpdateBallot                                PaxosAcceptor.java:302-322
  PISM.handleBatchedCommit (one committed slot per record)           PISM:1480-1528
  PISM.handleCommittedRequest                                        PISM:1432-1478
  PISM.extractExecuteAndCheckpoint                                   PISM:1619-1701
  PaxosAcceptor.putAndRemoveNextExecutable                           PaxosAcceptor.java:325-366
  PaxosAcceptor.reconstructDecision                                  PaxosAcceptor.java:369-385
  PaxosAcceptor.executed                                             PaxosAcceptor.java:462-474
  PaxosAcceptor.garbageCollectAccepted / garbageCollectDecisions     PaxosAcceptor.java:476-506
  Ballot.compareTo                                                   paxosutil/Ballot.java:60-73

`Acceptor` below is that reading (Java ints are small here, so plain Python ints stand for them; the
wraparound forms `a - b < 0` are kept as written).  `run_sequences` drives one group per op
sequence through gpx_accept_batch / gpx_commit_batch of a library behind the C-ABI (the oracle on the
CPU, the HIP engine on the GPU) and compares every reply, status, execution run and the final
HotRestoreInfo row with the model's.

An op is (kind, slot, bnum, median, stop):
  kind 'A'  an ACCEPT(ballot (bnum, COORD), slot, medianCheckpointedSlot = median, stop request or not)
  kind 'D'  a full DECISION (GPX_C_HASVALUE): handleCommittedRequest of a pvalue that has its value
  kind 'B'  one slot of a BATCHED_COMMIT(ballot, median): a real decision iff the stored ACCEPT has exactly
            that ballot, else the value-less placeholder of PISM:1510-1520
"""
import itertools

import numpy as np

from gigapaxos_amd import (Engine, hri_create, hri_initial, S_OK, S_STOPPED, R_TOLOG, R_STORED, A_STOP,
                           C_HASVALUE, C_STOP)

COORD = 100  # coordinator id of every ballot used here (ballots differ in their number only)
GET_ACCEPTED_PVALUES_FROM_DISK = True  # PaxosAcceptor.java:75-76 (the engine's GPX_F_ACCEPTS_FROM_DISK)


class I32(int):
    """a Java int: + and - wrap to 32 bits, comparisons are signed.  With slots of this type the reading below - which
    keeps every `a - b < 0` of the Java as written - computes what the Java computes at the int wrap too."""

    @staticmethod
    def _w(x):
        return I32(((int(x) + 2**31) % 2**32) - 2**31)

    def __add__(self, o):
        return I32._w(int(self) + int(o))

    __radd__ = __add__

    def __sub__(self, o):
        return I32._w(int(self) - int(o))

    def __rsub__(self, o):
        return I32._w(int(o) - int(self))

    def __neg__(self):
        return I32._w(-int(self))


def ballot_cmp(b1, b2):
    """Ballot.compareTo: by ballotNumber, then coordinatorID."""
    if b1[0] != b2[0]:
        return b1[0] - b2[0]
    return b1[1] - b2[1]


class PValue:
    __slots__ = ("ballot", "slot", "median", "has_value", "stop")

    def __init__(self, ballot, slot, median, has_value, stop):
        self.ballot, self.slot, self.median, self.has_value, self.stop = ballot, slot, median, has_value, stop


COVERAGE = {"runs": 0, "runs_of_2_or_more": 0, "rebuilt_from_accept": 0, "placeholders": 0, "stops_executed": 0,
            "dropped_stopped": 0, "nacks": 0, "gc_dropped_accepts": 0, "not_logged_repeat": 0,
            "kept_valued_decision": 0}


class Acceptor:
    """PaxosAcceptor + the PISM methods that drive it for ACCEPTs and commits."""

    def __init__(self, slot, ballot, gc_slot, from_disk=GET_ACCEPTED_PVALUES_FROM_DISK):
        self._slot = slot
        self.ballot = ballot           # (ballotNum, ballotCoord)
        self.acceptedGCSlot = gc_slot
        self.stopped = False
        self.accepted = {}             # acceptedProposals: slot -> PValue (the ACCEPT)
        self.committed = {}            # committedRequests: slot -> PValue (the decision)
        self.from_disk = from_disk

    # ---- PaxosAcceptor.java:476-494 ----------------------------------------------------------------
    def garbageCollectAccepted(self, gcSlot):
        if self._slot - gcSlot <= 0:
            gcSlot = self._slot - 1
        if gcSlot - self.acceptedGCSlot > 0:
            self.acceptedGCSlot = gcSlot
            for s in list(self.accepted):
                if s - gcSlot <= 0:
                    del self.accepted[s]
                    COVERAGE["gc_dropped_accepts"] += 1
        self.garbageCollectDecisions(gcSlot)

    # ---- PaxosAcceptor.java:496-506 ----------------------------------------------------------------
    def garbageCollectDecisions(self, slot):
        if slot - self._slot >= 0:
            return
        for s in list(self.committed):
            if slot - s > 0:
                del self.committed[s]

    # ---- PaxosAcceptor.java:302-322 ----------------------------------------------------------------
    def acceptAndUpdateBallot(self, accept):
        if self.stopped:
            return None
        if ballot_cmp(accept.ballot, self.ballot) >= 0:
            self.ballot = accept.ballot
            if accept.slot - self.acceptedGCSlot > 0:
                self.accepted[accept.slot] = accept
        self.garbageCollectAccepted(accept.median)
        return self.ballot

    # ---- PaxosAcceptor.java:369-385 ----------------------------------------------------------------
    def reconstructDecision(self, slot):
        rd = self.committed.get(slot)
        if rd is not None:
            if rd.has_value:
                return rd
            elif slot in self.accepted:
                if ballot_cmp(self.accepted[slot].ballot, rd.ballot) == 0:
                    a = self.accepted[slot]
                    COVERAGE["rebuilt_from_accept"] += 1
                    # new PValuePacket(accept).makeDecision(committed.getMedianCheckpointedSlot())
                    return PValue(a.ballot, a.slot, self.committed[slot].median, True, a.stop)
        return None

    # ---- PaxosAcceptor.java:462-474 ----------------------------------------------------------------
    def executed(self, s, stop):
        assert s == self._slot
        self._slot += 1
        if stop:
            self.stopped = True
            COVERAGE["stops_executed"] += 1
        if self.stopped:
            self.committed.clear()

    # ---- PaxosAcceptor.java:325-366 ----------------------------------------------------------------
    def putAndRemoveNextExecutable(self, decision):
        if self.stopped:
            return None
        self.garbageCollectAccepted(decision.median)
        if decision.slot - self._slot >= 0:
            if decision.slot not in self.committed or not self.committed[decision.slot].has_value:
                self.committed[decision.slot] = decision
            elif decision is not self.committed[decision.slot]:
                COVERAGE["kept_valued_decision"] += 1
        nextExecutable = None
        if self._slot in self.committed:
            nextExecutable = self.reconstructDecision(self._slot)
            if nextExecutable is not None and nextExecutable.has_value:
                del self.committed[self._slot]
                self.executed(nextExecutable.slot, nextExecutable.stop)
        if nextExecutable is not None and self.from_disk:
            self.accepted.pop(nextExecutable.slot, None)
        return nextExecutable

    # ---- PISM:1619-1701 (the app always executes; copyEpochFinalCheckpointState / logStop succeed) ----
    def extractExecuteAndCheckpoint(self, loggedDecision):
        """-> (first executed slot, number executed in order) or None"""
        if self.stopped:
            return None
        first, count = self._slot, 0
        while True:
            inorder = self.putAndRemoveNextExecutable(loggedDecision)
            if inorder is None:
                break
            count += 1
            if inorder.stop:
                break
        if count:
            COVERAGE["runs"] += 1
            COVERAGE["runs_of_2_or_more"] += count >= 2
        return (first, count) if count else None

    # ---- PISM:1080-1166 ---------------------------------------------------------------------------------
    def handleAccept(self, accept):
        """-> (status, r_bnum, r_bcoord, r_maxcp, r_flags, run)"""
        if self.stopped:                                   # PISM:456-460
            COVERAGE["dropped_stopped"] += 1
            return (S_STOPPED, 0, 0, 0, 0, None)
        prev = self.accepted.get(accept.slot)              # PISM:1123
        gc_before = self.acceptedGCSlot
        ballot_ok = ballot_cmp(accept.ballot, self.ballot) >= 0
        ballot = self.acceptAndUpdateBallot(accept)
        # (PISM.garbageCollectAccepted, PISM:1234, has an empty body)
        reply_maxcp = self._slot - 1                       # GC_MAJORITY_EXECUTED: getSlot() - 1
        toLog = (ballot_cmp(accept.ballot, ballot) >= 0 and accept.slot - self.acceptedGCSlot > 0 and
                 (prev is None or ballot_cmp(prev.ballot, accept.ballot) < 0))
        # the engine's GPX_R_STORED: "put into acceptedProposals" (PaxosAcceptor.java:315-316)
        stored = ballot_ok and accept.slot - gc_before > 0
        flags = (R_TOLOG if toLog else 0) | (R_STORED if stored else 0)
        COVERAGE["nacks"] += not ballot_ok
        COVERAGE["not_logged_repeat"] += ballot_ok and prev is not None and not toLog
        run = None
        rd = self.reconstructDecision(accept.slot)         # PISM:1158-1161
        if rd is not None:
            run = self.handleCommittedRequest(rd)
        return (S_OK, ballot[0], ballot[1], reply_maxcp, flags, run)

    # ---- PISM:1432-1478 ---------------------------------------------------------------------------------
    def handleCommittedRequest(self, committed):
        return self.extractExecuteAndCheckpoint(committed)

    def handleDecision(self, ballot, slot, median, stop):
        """a full DECISION packet (it has its request value)"""
        if self.stopped:                                   # PISM:456-460
            return (S_STOPPED, None)
        return (S_OK, self.handleCommittedRequest(PValue(ballot, slot, median, True, stop)))

    # ---- PISM:1480-1528, one committed slot ------------------------------------------------------------
    def handleBatchedCommitSlot(self, ballot, slot, median):
        if self.stopped:                                   # PISM:456-460
            return (S_STOPPED, None)
        accept = self.accepted.get(slot)
        if accept is not None and ballot_cmp(accept.ballot, ballot) == 0:
            d = PValue(accept.ballot, accept.slot, median, True, accept.stop)
        else:
            d = PValue(ballot, slot, median, False, False)  # placeholder: null request value
            COVERAGE["placeholders"] += 1
        return (S_OK, self.handleCommittedRequest(d))

    def apply(self, op):
        kind, slot, bnum, median, stop = op
        if kind == "A":
            return self.handleAccept(PValue((bnum, COORD), slot, median, True, bool(stop)))
        if kind == "D":
            return self.handleDecision((bnum, COORD), slot, median, bool(stop))
        return self.handleBatchedCommitSlot((bnum, COORD), slot, median)

    def row(self):
        return (self._slot, self.ballot[0], self.ballot[1], self.acceptedGCSlot)


def alphabet(slots, bnums, medians, stops=(0, 1)):
    ops = [("A", s, b, m, st) for s in slots for b in bnums for m in medians for st in stops]
    ops += [("D", s, b, m, st) for s in slots for b in bnums for m in medians for st in stops]
    ops += [("B", s, b, m, 0) for s in slots for b in bnums for m in medians]
    return ops


def _segments(seq):
    """a sequence as maximal runs of ops that go through the same call: [('A', [ops]), ('C', [ops]), ...]"""
    segs = []
    for op in seq:
        call = "A" if op[0] == "A" else "C"
        if segs and segs[-1][0] == call:
            segs[-1][1].append(op)
        else:
            segs.append((call, [op]))
    return segs


def run_sequences(lib, seqs, init="create", order="interleaved", from_disk=True, sample_dumps=64, promise=False, base=0):
    """One group per sequence.  Phase p of a group = its p-th run of same-call ops; phase p of all groups
    goes out as ONE batch per call type, a group's records in sequence order (so a group has several
    records per batch wherever its sequence repeats a call).  order = 'interleaved' (records of different
    groups alternate: the partition path) or 'grouped' (gidx non-decreasing: the direct path).
    Returns the number of records checked."""
    G = len(seqs)
    flags = 1 if from_disk else 0
    e = Engine(lib, COORD + 1, G, kmax=3, window=8, max_batch=max(1 << 16, 8 * G), flags=flags)
    members = np.tile(np.array([COORD, COORD + 1, COORD + 2], np.int32), (G, 1))
    rows = (hri_create if init == "create" else hri_initial)(G, 3, COORD)
    if base:
        # the same sequences with every slot (the ops', the medians', the instance's own) moved by `base`, Java ints
        # wrapping: the instance is restored (HotRestoreInfo) at slot 1 + base
        seqs = [tuple((k, I32(sl) + base, bn, I32(md) + base, st) for k, sl, bn, md, st in sq) for sq in seqs]
        rows["acc_slot"] = int(I32(1) + base)
        rows["acc_gc_slot"] = int(I32(-1 if init == "create" else 0) + base)
    assert (e.create_groups(np.arange(G), members, 3, rows) == S_OK).all()
    if promise:
        from gigapaxos_amd import ORDERED_ACCEPT, ORDERED_COMMIT
        e.set_ordered_batches(ORDERED_ACCEPT | ORDERED_COMMIT)
    gc0 = -1 if init == "create" else 0
    models = [Acceptor(I32(1) + base if base else 1, (0, COORD), I32(gc0) + base if base else gc0, from_disk) for _ in range(G)]
    segs = [_segments(s) for s in seqs]
    nphase = max(len(s) for s in segs)
    checked = 0
    for p in range(nphase):
        for call in ("A", "C"):
            recs = []  # (position in the group's run, group, op)
            for g, sg in enumerate(segs):
                if p < len(sg) and sg[p][0] == call:
                    recs += [(j, g, op) for j, op in enumerate(sg[p][1])]
            if not recs:
                continue
            recs.sort(key=(lambda r: (r[0], r[1])) if order == "interleaved" else (lambda r: (r[1], r[0])))
            n = len(recs)
            gi = np.array([r[1] for r in recs], np.int32)
            sl = np.array([r[2][1] for r in recs], np.int32)
            bn = np.array([r[2][2] for r in recs], np.int32)
            bc = np.full(n, COORD, np.int32)
            md = np.array([r[2][3] for r in recs], np.int32)
            # the model, record by record in batch order (a group's records keep their order)
            want = [models[r[1]].apply(r[2]) for r in recs]
            runs_by_group = {}
            for (j, g, op), w in zip(recs, want):
                run = w[-1]
                if run is not None:
                    runs_by_group.setdefault(g, []).append((g, run[0], run[1]))
            want_runs = [t for g in sorted(runs_by_group) for t in runs_by_group[g]]
            if call == "A":
                fl = np.array([A_STOP if r[2][4] else 0 for r in recs], np.uint8)
                (rb, rc, rm, rf, st), runs = e.accept(gi, bn, bc, sl, md, fl)
                got = list(zip(st.tolist(), rb.tolist(), rc.tolist(), rm.tolist(), rf.tolist()))
                exp = [w[:5] for w in want]
            else:
                kd = np.array([(C_HASVALUE | (C_STOP if r[2][4] else 0)) if r[2][0] == "D" else 0 for r in recs],
                              np.uint8)
                st, runs = e.commit(gi, bn, bc, sl, md, kd)
                got = st.tolist()
                exp = [w[0] for w in want]
            if got != exp:
                i = next(i for i in range(n) if got[i] != exp[i])
                raise AssertionError(f"phase {p} call {call} record {i}: sequence {seqs[recs[i][1]]} op {recs[i][2]}: "
                                     f"got {got[i]}, the Java gives {exp[i]}")
            got_runs = [tuple(t) for t in runs.as_tuple_array().tolist()]
            if got_runs != want_runs:
                bad = next((a, b) for a, b in itertools.zip_longest(got_runs, want_runs) if a != b)
                g = (bad[0] or bad[1])[0]
                raise AssertionError(f"phase {p} call {call}: execution runs differ at {bad}: sequence {seqs[g]}")
            checked += n
    snap, st = e.snapshot(np.arange(G))
    assert (st == S_OK).all()
    got_rows = list(zip(snap["acc_slot"].tolist(), snap["acc_bnum"].tolist(), snap["acc_bcoord"].tolist(),
                        snap["acc_gc_slot"].tolist()))
    exp_rows = [m.row() for m in models]
    if got_rows != exp_rows:
        g = next(g for g in range(G) if got_rows[g] != exp_rows[g])
        raise AssertionError(f"final acceptor row of sequence {seqs[g]}: got {got_rows[g]}, the Java gives {exp_rows[g]}")
    # full state of a sample: the maps themselves (layout: docs/HISTORY.md, state dump)
    rng = np.random.default_rng(G)
    for g in rng.choice(G, size=min(sample_dumps, G), replace=False).tolist():
        d = e.dump(g).tolist()
        m = models[g]
        k = d[2]
        pos = 3 + k
        a_slot, a_bn, a_bc, a_gc, stopped = d[pos:pos + 5]
        pos += 5
        na = d[pos]
        acc = [tuple(d[pos + 1 + 4 * i: pos + 5 + 4 * i]) for i in range(na)]
        pos += 1 + 4 * na
        nc = d[pos]
        com = [tuple(d[pos + 1 + 6 * i: pos + 7 + 6 * i]) for i in range(nc)]
        assert (a_slot, a_bn, a_bc, a_gc, stopped) == m.row() + (1 if m.stopped else 0,), (seqs[g], d)
        key = lambda t: int(I32(t[0]) - m._slot)   # noqa: E731  (the maps in order of distance from the next slot: at the wrap too)
        if not base:
            assert acc == sorted(acc) and com == sorted(com)   # (the dump lists them by slot)
        assert sorted(acc, key=key) == sorted([(s, v.ballot[0], v.ballot[1], 1 if v.stop else 0) for s, v in m.accepted.items()], key=key), (seqs[g], acc)
        assert sorted(com, key=key) == sorted([(s, v.ballot[0], v.ballot[1], v.median, 1 if v.has_value else 0, 1 if (v.stop and v.has_value) else 0)
                                               for s, v in m.committed.items()], key=key), (seqs[g], com)
    e.close()
    return checked


# ---- the enumeration plan (bounded exhaustive + seeded longer sequences) ---------------------------
WIDE = alphabet(slots=(0, 1, 2, 3), bnums=(0, 1, 2), medians=(-1, 0, 1, 2, 3))        # 360 ops
MEDIUM = alphabet(slots=(1, 2, 3), bnums=(0, 1, 2), medians=(-1, 2))                   # 90 ops
SMALL = alphabet(slots=(1, 2), bnums=(0, 1), medians=(-1, 1), stops=(0,)) + [("A", 1, 0, -1, 1), ("D", 2, 1, 1, 1)]  # 22 ops


def plan(scale=1.0, seed=2024):
    """[(name, sequences)]: every sequence of length <= 2 over WIDE, of length 3 over MEDIUM (`scale` < 1
    keeps a seeded fraction), of length 4 over SMALL, and seeded random sequences of length 5 and 6 over
    WIDE."""
    rng = np.random.default_rng(seed)
    out = []
    out.append(("len1-wide", [(a,) for a in WIDE]))
    out.append(("len2-wide", list(itertools.product(WIDE, repeat=2))))
    l3 = list(itertools.product(MEDIUM, repeat=3))
    if scale < 1.0:
        keep = rng.random(len(l3)) < scale
        l3 = [s for s, k in zip(l3, keep) if k]
    out.append(("len3-medium", l3))
    l4 = list(itertools.product(SMALL, repeat=4))
    if scale < 1.0:
        keep = rng.random(len(l4)) < scale
        l4 = [s for s, k in zip(l4, keep) if k]
    out.append(("len4-small", l4))
    nrand = int(120_000 * scale)
    for L in (5, 6):
        idx = rng.integers(0, len(WIDE), (nrand, L))
        out.append((f"len{L}-wide-random", [tuple(WIDE[i] for i in row) for row in idx.tolist()]))
    return out


def run_long_random(lib, n, lengths=(8, 12), seed=77, orders=("interleaved", "grouped"), inits=("create", "initial")):
    """Seeded random op sequences far longer than the exhaustive plans reach (a group lives through many
    accept / commit / placeholder / stop interleavings): n sequences per length."""
    rng = np.random.default_rng(seed)
    total = 0
    for L in lengths:
        idx = rng.integers(0, len(WIDE), (n, L))
        seqs = [tuple(WIDE[i] for i in row) for row in idx.tolist()]
        for i, order in enumerate(orders):
            total += run_sequences(lib, seqs, init=inits[i % len(inits)], order=order)
    return total


def run_plan(lib, scale=1.0, orders=("interleaved", "grouped"), inits=("create", "initial"), from_disk=(True,),
             promise=False, skip=(), both_inits_below=200_000, single_order_above=None):
    total = 0
    for p, (name, seqs) in enumerate(plan(scale)):
        if name in skip:
            continue
        plan_orders = orders
        if single_order_above is not None and len(seqs) > single_order_above and len(orders) > 1:
            plan_orders = (orders[p % len(orders)],)  # the exhaustive plans: one batch order each, alternating
        for i, order in enumerate(plan_orders):
            # every batch order sees both initial rows on the short plans; the long ones alternate
            for init in (inits if len(seqs) < both_inits_below else (inits[i % len(inits)],)):
                for fd in from_disk:
                    total += run_sequences(lib, seqs, init=init, order=order, from_disk=fd, promise=promise)
    return total
