"""Coordinator side of the view change, pinned on the oracle (CPU).

The reference's only executable statement of this phase is PaxosCoordinatorState.main
(PaxosCoordinatorState.java:1008-1180); its prepare phase is restated here step by step with a
9-member group.  Its tail assertions (`slot >= maxMinSlot`, "stops are only followed by stops") do
not hold for its own inputs - with assertions enabled processStop's `assert (false)` fires for the
pre-active stop at slot 1 followed by carried requests in the same (re-stamped) ballot - so the
expected proposal list below is what the code computes with assertions disabled (production)."""
from tests.election_common import (pcs_main_scenario, small_scenarios, boundary_scenario, fuzz_run, S_PREACTIVE, S_WINDOW,
                                   EB_PREPARING, EB_ACTIVE, EB_RESEND, EB_UNCHANGED, V_IGNORED, V_RECORDED,
                                   V_ELECTED, V_PREEMPTED, E_CARRY, E_NOOP, E_PREACTIVE, E_NEWSTOP, PV_STOP)
from gigapaxos_amd import S_OK, S_REFUSED


def test_pcs_main_prepare_phase(oracle_lib):
    t = pcs_main_scenario(oracle_lib)
    it = iter(t)
    assert next(it) == ("begin", [EB_PREPARING])
    # pre-active propose() returns no ACCEPT; slots from the acceptor's slot 0; ballot (2, 21)
    assert next(it) == ("propose", 0, 2, 21, 0, S_PREACTIVE)
    assert next(it) == ("propose", 1, 2, 21, 0, S_PREACTIVE)
    assert next(it)[-1] == S_REFUSED  # after the stop
    # canIgnorePrepareReply: lower ballot number, lower coordinator id, non-member
    # running for coordinator: the PREPARE (2, 21) from slot 0 is what a timer would resend
    assert next(it) == ("poke", [[2], [0], [2], [21], [0], [0], [0], [0]])
    for _ in range(3):
        r = next(it)
        assert r[2] == V_IGNORED and r[5] == []
    r = next(it)
    assert (r[1], r[2]) == (30, V_RECORDED)          # members[2]: first legitimate reply
    assert next(it)[2] == V_IGNORED                  # no duplicates
    assert next(it)[1:3] == (21, V_RECORDED)         # members[0], the self reply
    assert next(it)[2] == V_IGNORED
    assert next(it)[1:3] == (40, V_RECORDED)         # members[4]
    assert next(it)[2] == V_IGNORED
    kinds = [next(it) for _ in range(5)]             # members 0, 2, 4, 6, 8
    assert [x[2] for x in kinds] == [V_IGNORED, V_IGNORED, V_IGNORED, V_RECORDED, V_ELECTED]
    el = kinds[-1]
    assert el[3] == 0 and el[4] == S_OK              # median of the recorded min slots
    assert el[5] == [
        (0, E_PREACTIVE, 100, 0),       # reqs[0]: its carried-over copy at slot 2 was overwritten
        (1, E_PREACTIVE, 900, PV_STOP),
        (2, E_CARRY, 101, 0),           # slot 2 in the higher ballot (1, 21) won
        (3, E_NOOP, 0, 0), (4, E_NOOP, 0, 0), (5, E_NOOP, 0, 0),
        (6, E_CARRY, 102, 0), (7, E_CARRY, 103, 0), (8, E_CARRY, 104, 0), (9, E_CARRY, 105, 0),
        (10, E_NEWSTOP, 0, PV_STOP),    # processStop: a stop exists and the last proposal is none
    ]
    next(it)  # dump
    # active: the head-of-line proposal (the acceptor's slot 0) is the one ACCEPT a poke re-sends,
    # with the median of the node slots as they are now and nobody heard from yet
    assert next(it) == ("poke", [[1], [0], [2], [21], [0], [0], [0], [0]])
    assert next(it)[-1] == S_REFUSED


def test_small_scenarios(oracle_lib):
    t = dict((x[0], x[1:]) for x in small_scenarios(oracle_lib))
    assert t["begin"][0] == [EB_ACTIVE] + [EB_PREPARING] * 7
    assert t["begin-again"][0] == [EB_UNCHANGED, EB_RESEND, EB_UNCHANGED, EB_PREPARING] + [EB_RESEND] * 4
    vk, em, st, lists = t["preempted"]
    assert vk == [V_PREEMPTED] and lists == [[(5, E_PREACTIVE, 11, 0), (6, E_PREACTIVE, 12, PV_STOP)]]
    vk, em, st, lists = t["elected-empty"]
    assert vk == [V_RECORDED, V_ELECTED] and lists == [[], []] and em == [0, 4]  # node slots [4, 4, -1]: median 4
    vk, em, st, lists = t["elected3"]
    assert vk == [V_RECORDED, V_ELECTED]
    # maxMin = 7 (min slots 6 -> min(6, 7..9) = 6 for node 2, 7 for node 0), carried 7, 8, 9; the
    # pre-active at slot 7 (handle 33) loses its slot to the carried value and, like 5 and 6 (31 and
    # the duplicate 32), is re-proposed after slot 9 - isDuplicate only guards a pre-active that
    # keeps its slot inside the carried range (PCS:419-427)
    assert lists[1] == [(7, E_CARRY, 77, 0), (8, E_CARRY, 88, 0), (9, E_CARRY, 32, 0),
                        (10, E_PREACTIVE, 31, 0), (11, E_PREACTIVE, 32, 0), (12, E_PREACTIVE, 33, 0)]
    assert t["ar4"][0] == []
    vk, em, st, lists = t["elected5"]
    assert vk == [V_RECORDED, V_ELECTED]
    assert lists[1] == [(5, E_CARRY, 51, PV_STOP), (6, E_CARRY, 52, 0), (7, E_NEWSTOP, 0, PV_STOP)]
    assert t["dup"][0] is True
    vk, em, st, lists = t["window6"]
    # reply 0: slots 5 and 13 collide in a ring of 8 -> dropped whole; reply 2 is recorded and makes a
    # majority, but slots 5..14 do not fit 8 entries -> recorded, not elected
    assert st == [S_WINDOW, S_OK, S_WINDOW] and vk == [V_IGNORED, V_RECORDED, V_RECORDED]


def test_fuzz_runs_are_deterministic(oracle_lib):
    a = fuzz_run(oracle_lib, 5, G=48, steps=40)
    b = fuzz_run(oracle_lib, 5, G=48, steps=40)
    assert a == b
    kinds = [v for x in a if x[0] == "reply" for v in x[1]]
    assert V_ELECTED in kinds and V_PREEMPTED in kinds and V_RECORDED in kinds


def test_failover_end_to_end(oracle_lib):
    """Node 0 dies with ACCEPTs in flight; node 1 runs, is elected and every accepted value is decided
    at its own slot (invariants asserted inside failover_run)."""
    from tests.failover_common import failover_run

    t = failover_run(oracle_lib, G=120, seed=3)
    assert len(t["inflight"]) > 0
    kinds = {e[1] for lists in t["reply2"][2] for e in lists}
    assert {1, 2, 3} <= kinds  # carried over, no-op filled and pre-active entries all occur


def test_half_range_boundary_is_refused(oracle_lib):
    t = dict((x[0], x[1:]) for x in boundary_scenario(oracle_lib))
    vk, em, st, lists = t["replies"]
    # group 0: recorded twice, the majority reply cannot complete the view change (GPX_S_WINDOW);
    # group 1 (member 0 answered): elected, slot Integer.MAX_VALUE carried over
    assert vk == [V_RECORDED, V_RECORDED, V_RECORDED, V_ELECTED] and st == [S_OK, S_WINDOW, S_OK, S_OK]
    assert lists[3] == [(2 ** 31 - 1, E_CARRY, 8, 0)]


def test_election_begin_sequences_against_java_reading(oracle_lib):
    """PaxosCoordinator.makeCoordinator (PaxosCoordinator.java:66-89) read on its own: a coordinator that is missing
    or has a LOWER ballot is replaced by a fresh PaxosCoordinatorState (active at once iff the ballot number is 0,
    else it prepares); the SAME ballot, not yet active: the PREPARE is sent again; anything else: nothing.  Random
    sequences of election_begin calls with ballot numbers 0..3 and completed elections in between, per group, against
    that reading: the status of every call, which groups have an active coordinator (HotRestoreInfo), and what
    gpx_poke_scan says is waiting (the PREPARE of a coordinator that is not active)."""
    import numpy as np
    from gigapaxos_amd import Engine, hri_create
    me, members = 101, [100, 101, 102]
    rng = np.random.default_rng(17)
    G = 3000
    e = Engine(oracle_lib, me, G, kmax=3, window=8, max_batch=1 << 14)
    coord0 = rng.choice(members, G).astype(np.int32)
    assert (e.create_groups(np.arange(G), np.tile(np.array(members, np.int32), (G, 1)), 3, hri_create(G, 3, coord0)) == S_OK).all()
    c = [((0, me), True) if coord0[g] == me else None for g in range(G)]       # hotRestore: a coordinator only where it is me
    seen = set()
    for step in range(12):
        gs = np.nonzero(rng.random(G) < 0.6)[0].astype(np.int32)
        bn = rng.integers(0, 4, gs.shape[0]).astype(np.int32)
        st = e.election_begin(gs, bn)
        for i, g in enumerate(gs.tolist()):
            new = (int(bn[i]), me)
            if c[g] is None or c[g][0] < new:
                c[g] = (new, new[0] == 0)
                want = EB_ACTIVE if new[0] == 0 else EB_PREPARING
            elif c[g][0] == new and not c[g][1]:
                want = EB_RESEND
            else:
                want = EB_UNCHANGED
            assert int(st[i]) == want, (step, g, new, c[g])
            seen.add(want)
        # some of the running elections complete: two members answer in the candidate's ballot
        run = np.array([g for g in range(G) if c[g] is not None and not c[g][1] and rng.random() < 0.5], np.int32)
        for k, acceptor in enumerate((100, 102)):
            if run.shape[0] == 0:
                break
            rb = np.array([c[g][0][0] for g in run.tolist()], np.int32)
            (vk, em, rst), lists = e.prepare_reply(run, np.full(run.shape[0], acceptor, np.int32), rb, np.full(run.shape[0], me, np.int32),
                                                   np.ones(run.shape[0], np.int32))
            assert (rst == S_OK).all() and (vk == (V_RECORDED if k == 0 else V_ELECTED)).all()
        for g in run.tolist():
            c[g] = (c[g][0], True)
        snap, _ = e.snapshot(np.arange(G))
        active = np.array([x is not None and x[1] for x in c])
        assert ((snap["has_coord"] != 0) == active).all()
        assert (snap["coord_bnum"][active] == np.array([x[0][0] for x in c if x is not None and x[1]], np.int32)).all()
        pk, sl, pbn, pbc, md, fl, hd, pst = e.poke_scan()
        waiting = np.array([x is not None and not x[1] for x in c])
        assert ((pk == 2) == waiting).all() and (pk[~waiting] == 0).all()         # GPX_POKE_PREPARE / GPX_POKE_NONE
        assert (pbn[waiting] == np.array([x[0][0] for x in c if x is not None and not x[1]], np.int32)).all() and (pbc[waiting] == me).all()
    assert seen == {EB_PREPARING, EB_ACTIVE, EB_RESEND, EB_UNCHANGED}
    e.close()
