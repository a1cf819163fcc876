"""Pins the oracle's restatement of RequestBatcher / roundRobinCoordinator / gap detection
 against hand-computed answers.  CPU only."""
import numpy as np

from gigapaxos_amd import wire as W
from tests import host_rows_common as H


def test_request_batcher_known_answer(oracle_lib):
    H.request_batch_kat(oracle_lib)


def test_request_batcher_invariants(oracle_lib):
    (leader, status, b), _ = H.request_batch_run(oracle_lib, 3)
    leader, b = np.array(leader), {k: np.array(v) for k, v in b.items()}
    assert b["count"].sum() == (np.array(status) == 0).sum()      # every queued request is in a batch
    assert (np.diff(b["gidx"]) >= 0).all()                          # grouped by gidx
    assert (b["bytes"][b["count"] > 1] <= 2000).all() and (b["size"][b["count"] > 1] <= 400).all()
    assert (leader[b["leader"]] == b["leader"]).all()               # heads lead themselves


def test_round_robin_coordinator(oracle_lib):
    """members[Math.abs(ballotnum + paxosID.hashCode()) % k] (PISM:2251-2256), Java int arithmetic."""
    out, names, members, ks = H.coordinator_run(oracle_lib)
    for bi, bal in enumerate((0, 1, 7, -5, 2**31 - 1)):
        for qi, g in enumerate(range(-1, 501)):
            want = -2**31
            if 0 <= g < 480 and g % 9 != 4:
                x = (bal + W.java_string_hash(names[g]) + 2**31) % 2**32 - 2**31
                ax = -x if x < 0 else x
                ax = ax if ax < 2**31 else -2**31
                if ax >= 0:
                    want = int(members[g, ax % int(ks[g])])
            assert out[bi][qi] == want, (bal, g)
    assert out[5] == [-2**31]


def test_gap_detection_known_answer(oracle_lib):
    from gigapaxos_amd import Engine, hri_create, C_HASVALUE
    e = Engine(oracle_lib, 100, 4, kmax=3, window=16)
    we = W.WireEngine(e)
    mem = np.tile(np.array([100, 101, 102], np.int32), (3, 1))
    e.create_groups(np.arange(3), mem, 3, hri_create(3, 3, 100))
    z = lambda n: np.zeros(n, np.int32)  # noqa: E731
    c = lambda n: np.full(n, 100, np.int32)  # noqa: E731
    # group 0: accept for slot 3; decisions 2 (value), 3 (meta, accept present), 5 (meta, no accept), 7 (value)
    e.accept([0], z(1), c(1), [3], z(1))
    e.commit([0, 0, 0, 0], z(4), c(4), [2, 3, 5, 7], z(4), np.array([C_HASVALUE, 0, 0, C_HASVALUE], np.uint8))
    # group 1: executes 1..2, nothing pending
    e.commit([1, 1], z(2), c(2), [1, 2], z(2), np.array([C_HASVALUE, C_HASVALUE], np.uint8))
    first, maxc, missing, sync, st = W.gap_scan(we, [0, 1, 2, 3], threshold=5)
    assert first.tolist() == [1, 3, 1, 0] and maxc.tolist() == [7, 2, 0, 0] and st.tolist() == [0, 0, 0, 1]
    # slots 1..6 examined (i < maxCommittedSlot): 1 missing, 2 ok, 3 ok (meta + accept), 4, 5 (meta, no accept), 6
    assert missing.tolist() == [0b111001, 0, 0, 0]
    assert sync.tolist() == [1, 0, 0, 0]          # gap 6 >= 5; group 1: gap -1
    first, maxc, missing, sync, st = W.gap_scan(we, [0, 1, 2], threshold=100, size_limit=3)
    assert missing.tolist() == [0b001, 0, 0] and sync.tolist() == [1, 0, 0]  # expectedSlot 1 and gap >= 100/100
    assert W.gap_scan(we, [1], 100, W.SYNC_FORCE)[3].tolist() == [1]
    e.close()


def test_election_scan_known_answer(oracle_lib):
    """PISM.checkRunForCoordinator (PISM:2090-2176): run iff I hold no coordinator at or above my
    acceptor's ballot AND (its coordinator is me, or it is down and I am the next member after it, or
    it is long dead); the PREPARE carries ballot (b + 1, me) and firstUndecidedSlot = my slot."""
    from gigapaxos_amd import Engine, hri_create
    e = Engine(oracle_lib, 101, 8, kmax=3, window=8)
    we = W.WireEngine(e)
    mem = np.array([[100, 101, 102]] * 4 + [[100, 102, 103]] + [[101, 0, 0]], np.int32)
    ks = np.array([3, 3, 3, 3, 3, 1], np.uint8)
    #                        group: 0    1    2    3    4    5
    rows = hri_create(6, 3, np.array([100, 101, 102, 100, 100, 101], np.int32))
    e.create_groups(np.arange(6), mem, ks, rows)
    # group 3: the acceptor has promised (2, 102); group 1: I coordinate it already
    e.prepare([3], [2], [102], [1])
    run, pb, pf, st = W.election_scan(we, [0, 1, 2, 3, 4, 5, 6], down_nodes=[100])
    #  0: coordinator 100 down, next after 100 is 101 = me -> NEXT, PREPARE (1, 101) from slot 1
    #  1: I am the coordinator and hold it -> no        2: coordinator 102 is up -> no
    #  3: ballot (2,102), 102 up -> no                  4: 100 down but next after 100 is 102 -> no
    #  5: single member group, mine, held -> no         6: no such group
    assert run.tolist() == [2, 0, 0, 0, 0, 0, 0] and st.tolist() == [0, 0, 0, 0, 0, 0, 1]
    assert pb.tolist()[:1] == [1] and pf.tolist()[:1] == [1]
    run, pb, pf, st = W.election_scan(we, [0, 2, 3, 4], down_nodes=[100, 102], long_dead_nodes=[100])
    assert run.tolist() == [2, 0, 0, 3]            # 2 / 3: next after 102 is 100, not me; 4: long dead
    assert pb.tolist() == [1, 0, 0, 1]
    assert W.election_scan(we, [3], down_nodes=[102], long_dead_nodes=[102])[0].tolist() == [3]
    assert W.election_scan(we, [3], down_nodes=[102], long_dead_nodes=[102])[1].tolist() == [3]   # (2 + 1, me)
    assert W.election_scan(we, [1, 2], force=True)[0].tolist() == [4, 4]
    e.close()


def dequeue_all(g, est, weight, stop, G, max_bytes, max_size):
    """RequestBatcher.enqueueImpl for every request in arrival order, then dequeueImpl until the map is empty
    (RequestBatcher.java:111-129, 163-239), read from the Java: the head of the first list is always plucked; the
    followers while `(totalByteLength += next.lengthEstimate()) > limit || (totalBatchSize += next.batchSize() + 1)
    > MAX_BATCH_SIZE` does not hold (short-circuit: the size is not added when the bytes already broke the loop).
    The map's iteration order (which group is first) is the HashMap's: the batches are listed by group here."""
    queues, status = {}, []
    for i in range(len(g)):
        ok = 0 <= g[i] < G
        status.append(0 if ok else 1)                 # GPX_S_NOGROUP: no such row
        if ok:
            queues.setdefault(int(g[i]), []).append(i)
    leader = [-1] * len(g)
    batches = []
    for grp in sorted(queues):
        q = queues[grp]
        while q:
            first = q.pop(0)
            batch = [first]
            total_bytes, total_size = int(est[first]), int(weight[first])
            while q:
                nxt = q[0]
                total_bytes += int(est[nxt])
                if total_bytes > max_bytes:
                    break
                total_size += int(weight[nxt])
                if total_size > max_size:
                    break
                batch.append(q.pop(0))
            for i in batch:
                leader[i] = first
            batches.append((grp, first, len(batch), sum(int(est[i]) for i in batch), sum(int(weight[i]) for i in batch),
                            int(any(stop[i] for i in batch))))
    return leader, status, batches


def test_request_batcher_random_bursts_against_java_reading(oracle_lib):
    from gigapaxos_amd import Engine
    for seed, n, G, limits in ((1, 20_000, 3000, (2000, 400)), (2, 30_000, 50, (900, 7)), (3, 5_000, 5000, (1 << 20, 2000)),
                               (4, 20_000, 300, (400, 1))):
        rng = np.random.default_rng(seed)
        e = Engine(oracle_lib, 100, G, kmax=3, window=8, max_batch=1 << 16)
        we = W.WireEngine(e)
        g = rng.integers(-1, G + 1, n).astype(np.int32)
        g[rng.random(n) < 0.2] = 17                   # one group with thousands of queued requests
        est = rng.integers(1, 400, n).astype(np.int32)
        wt = rng.choice([1, 1, 1, 2, 7, 300], size=n).astype(np.int32)   # batchSize() + 1 of an already batched request
        stop = (rng.random(n) < 0.01).astype(np.uint8)
        leader, status, b = W.request_batch(we, g, est, wt, stop, max_bytes=limits[0], max_size=limits[1])
        want_leader, want_status, want = dequeue_all(g, est, wt, stop, G, *limits)
        assert status.tolist() == want_status and leader.tolist() == want_leader, f"seed {seed}"
        got = list(zip(*[b[k].tolist() for k in ("gidx", "leader", "count", "bytes", "size", "stop")]))
        assert got == want, f"seed {seed}: {len(got)} batches against {len(want)}"
        e.close()


def test_election_scan_random_groups_against_java_reading(oracle_lib):
    """PISM.checkRunForCoordinator's condition (PISM:2090-2176) evaluated in Python for groups of one to five members
    whose acceptor ballots PREPAREs have moved to member and non-member coordinators, under several (down, long dead,
    forceRun) inputs: `!PaxosCoordinator.exists(coordinator, curBallot) && (curBallot.coordinatorID == myID ||
    (!isNodeUp(c) && (myID == getNextCoordinator(c, ..) || lastCoordinatorLongDead))) || forceRun`; the reason
    reported is the first disjunct that holds, in the Java's order; the PREPARE is (ballotNumber + 1, myID) from
    paxosState.getSlot().  (getNextCoordinator of a non-member is random in the Java: nobody is next here.)"""
    from gigapaxos_amd import Engine, hri_create, S_OK
    me = 102
    for seed in (1, 2, 3):
        rng = np.random.default_rng(seed)
        G, k = 1500, 5
        e = Engine(oracle_lib, me, G, kmax=k, window=8, max_batch=1 << 14)
        we = W.WireEngine(e)
        ks = rng.integers(1, k + 1, G).astype(np.uint8)
        members = np.zeros((G, k), np.int32)
        for g in range(G):
            members[g, :ks[g]] = np.sort(np.concatenate([[me], rng.choice([100, 101, 103, 104, 105, 106], size=ks[g] - 1,
                                                                         replace=False)]))
        coord0 = np.array([members[g, rng.integers(0, ks[g])] for g in range(G)], np.int32)
        created = np.arange(G - 40, dtype=np.int32)
        assert (e.create_groups(created, members[created], ks[created], hri_create(G - 40, k, coord0[created])) == S_OK).all()
        ballot = {g: (0, int(coord0[g])) for g in created.tolist()}          # the acceptor's
        mine = {g: (0, me) for g in created.tolist() if coord0[g] == me}     # my coordinator (hotRestore: only if it is me)
        n = 1500
        pg = rng.integers(0, G, n).astype(np.int32)
        bn = rng.integers(0, 4, n).astype(np.int32)
        bc = rng.choice([100, 101, 102, 103, 104, 105, 106, 999], size=n).astype(np.int32)
        e.prepare(pg, bn, bc, np.ones(n, np.int32))
        for g, b, c in zip(pg.tolist(), bn.tolist(), bc.tolist()):
            if g in ballot and (b, c) > ballot[g]:
                ballot[g] = (b, c)
        q = np.arange(-1, G + 1, dtype=np.int32)
        reasons = set()
        for down, longdead, force in (((), (), False), ((100,), (), False), ((100, 104), (104,), False), ((999, 106), (999,), False),
                                      ((100, 101, 103, 104, 105, 106, 999), (999, 105), False), ((), (100,), False), ((101,), (), True)):
            run, pb, pf, st = W.election_scan(we, q, down, longdead, force)
            for i, g in enumerate(q.tolist()):
                if g not in ballot:
                    assert (int(st[i]), int(run[i])) == (1, 0), (seed, g)
                    continue
                b, c = ballot[g]
                mem = members[g, :ks[g]].tolist()
                exists = g in mine and mine[g] >= (b, c)
                nxt = mem[(mem.index(c) + 1) % len(mem)] if c in mem else None
                why = 0
                if not exists and c == me:
                    why = 1
                elif not exists and c in down and nxt == me:
                    why = 2
                elif not exists and c in down and c in longdead:
                    why = 3
                elif force:
                    why = 4
                want = (0, why, b + 1, 1) if why else (0, 0, 0, 0)
                assert (int(st[i]), int(run[i]), int(pb[i]), int(pf[i])) == want, (seed, g, down, longdead, force)
                reasons.add(why)
        assert reasons == {0, 1, 2, 3, 4}
        e.close()
