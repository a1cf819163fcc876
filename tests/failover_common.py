"""Coordinator failover end to end over three engines (test driver, runs unchanged over the HIP library
and the oracle): ordinary rounds under node 0, node 0 dies with ACCEPTs in flight, node 1 finds the
groups it must run for (gpx_election_scan), runs (gpx_election_begin), takes client requests while
not yet active (gpx_propose_batch_h), PREPAREs nodes 1 and 2 (gpx_prepare_batch), feeds the replies
back (gpx_prepare_reply_batch), multicasts the ACCEPTs of the view change and goes on deciding.

Paths: PISM.checkRunForCoordinator :2090-2176, handlePrepare :900-1006, handlePrepareReply :1008-1068,
handleAccept :1070-1160, handleAcceptReply :1248-1364."""
import numpy as np

from gigapaxos_amd import S_OK, D_DECISION, A_STOP, C_HASVALUE
from gigapaxos_amd import wire as W
from gigapaxos_amd.loopback import LoopbackCluster
from tests.election_common import (S_PREACTIVE, EB_PREPARING, V_ELECTED, V_RECORDED, E_CARRY, E_NOOP, E_PREACTIVE,
                                   PV_STOP)

NOOP_HANDLE = 0


def failover_run(lib, G=200, seed=0, rounds_before=3, rounds_after=3, window=8):
    rng = np.random.default_rng(seed)
    ids = [0, 1, 2]
    cl = LoopbackCluster(lib, ids, G, window=window, max_batch=max(4096, 8 * G))
    e0, e1, e2 = (cl.engines[i] for i in ids)
    allg = np.arange(G, dtype=np.int32)
    value = {}   # (g, slot) -> handle of the value decided there
    hctr = [1]

    def fresh(n):
        h = np.arange(hctr[0], hctr[0] + n, dtype=np.int64)
        hctr[0] += n
        return h

    trace = {}
    for _ in range(rounds_before):
        dec = cl.round(allg)
        for g, s in dec[dec[:, 5] == D_DECISION][:, :2]:
            value[(int(g), int(s))] = int(fresh(1)[0])
    # --- in flight when node 0 dies: two proposals per group of a subset; each ACCEPT reaches a
    # random subset of the survivors, no reply is ever processed
    sub = allg[rng.random(G) < 0.7]
    inflight = {}
    for rep in range(2):
        slot, bn, bc, med, st = e0.propose(sub)
        assert (st == S_OK).all()
        h = fresh(sub.size)
        e0.accept(sub, bn, bc, slot, med)
        for eng, nid in ((e1, 1), (e2, 2)):
            got = rng.random(sub.size) < 0.5
            (rb, rc, rm, rf, ast), runs = eng.accept(sub[got], bn[got], bc[got], slot[got], med[got])
            assert (ast == S_OK).all() and runs.gidx.size == 0
            for g, s, hh in zip(sub[got], slot[got], h[got]):
                inflight[(int(g), int(s))] = int(hh)
    trace["inflight"] = sorted(inflight.items())
    # --- node 1 notices that node 0 is down
    we1 = W.WireEngine(e1)
    run, p_bnum, p_first, st = W.election_scan(we1, None, down_nodes=[0])
    assert (st == S_OK).all()
    trace["scan"] = (run.tolist(), p_bnum.tolist(), p_first.tolist())
    run2, _, _, _ = W.election_scan(W.WireEngine(e2), None, down_nodes=[0])
    assert not run2.any()  # node 2 is not next in line
    rg = allg[run != 0]
    assert rg.size == G
    es = e1.election_begin(rg, p_bnum[run != 0])
    assert (es == EB_PREPARING).all()
    # client requests reach node 1 before it is elected: pre-active proposals
    pre = allg[rng.random(G) < 0.4]
    hp = fresh(pre.size)
    slot, bn, bc, med, st = e1.propose(pre, handle=hp)
    assert (st == S_PREACTIVE).all()
    trace["preactive"] = (pre.tolist(), slot.tolist())
    # --- PREPARE to the survivors, replies back to node 1
    nb = p_bnum[rg]
    elected_lists = {}
    med_of = {}
    for eng, nid in ((e1, 1), (e2, 2)):
        (rb, rc, rgc, rf, pst), rows = eng.prepare(rg, nb, np.full(rg.size, 1, np.int32), p_first[rg])
        assert (pst == S_OK).all() and not (rf & 1).any()  # no NACK
        pvs = [[] for _ in range(rg.size)]
        for i, s, b, c in rows:
            pvs[i].append((s, b, c, inflight[(int(rg[i]), s)], 0))
        (vk, em, rst), lists = e1.prepare_reply(rg, np.full(rg.size, nid, np.int32), rb, rc, rgc + 1, pvs)
        assert (rst == S_OK).all()
        trace["reply%d" % nid] = (vk.tolist(), em.tolist(), lists)
        assert (vk == (V_RECORDED if nid == 1 else V_ELECTED)).all()
        if nid == 2:
            for i, g in enumerate(rg):
                elected_lists[int(g)] = lists[i]
                med_of[int(g)] = int(em[i])
    # --- the ACCEPTs of the view change, all in node 1's new ballot
    ag, aslot, aflag, amed = [], [], [], []
    for g in range(G):
        for s, kind, h, fl in elected_lists[g]:
            ag.append(g), aslot.append(s), aflag.append(A_STOP if fl & PV_STOP else 0), amed.append(med_of[g])
            want = inflight.get((g, s))
            if kind == E_CARRY:
                assert h == want  # safety: an accepted value is re-proposed at its own slot
            else:
                assert want is None
            value[(g, s)] = h if kind != E_NOOP else NOOP_HANDLE
    ag, aslot, amed = (np.asarray(x, np.int32) for x in (ag, aslot, amed))
    aflag = np.asarray(aflag, np.uint8)
    abn, abc = p_bnum[ag], np.full(ag.size, 1, np.int32)
    cl.coordinator[:] = 1
    decs = []
    votes = []
    for eng, nid in ((e1, 1), (e2, 2)):
        (rb, rc, rm, rf, ast), runs = eng.accept(ag, abn, abc, aslot, amed, aflag)
        assert (ast == S_OK).all()
        votes.append((ag, rb, rc, aslot, np.full(ag.size, nid, np.int32), rm))
    for v in votes:
        decs.append(e1.accept_reply(*v).as_tuple_array())
    dec = np.concatenate(decs)
    assert (dec[:, 5] == D_DECISION).all() and dec.shape[0] == ag.size
    trace["view_change_decisions"] = dec.tolist()
    for eng, nid in ((e1, 1), (e2, 2)):
        st2, runs = eng.commit(dec[:, 0], dec[:, 2], dec[:, 3], dec[:, 1], dec[:, 4],
                               np.full(dec.shape[0], C_HASVALUE, np.uint8))
        assert (st2 == S_OK).all()
        cl._log_runs(nid, runs)
    # --- life goes on under node 1 (node 0 stays down: deliver to 1 and 2 only)
    cl.node_ids = [1, 2]
    for _ in range(rounds_after):
        dec = cl.round(allg)
        assert (dec[:, 5] == D_DECISION).all() and dec.shape[0] == G
        for g, s in dec[:, :2]:
            value[(int(g), int(s))] = int(fresh(1)[0])
    cl.node_ids = ids
    ex = {nid: cl.executed(nid) for nid in (1, 2)}
    trace["executed"] = {nid: ex[nid].tolist() for nid in ex}
    trace["dumps"] = [[cl.engines[n].dump(g).tolist() for g in range(0, G, max(1, G // 16))] for n in (1, 2)]
    # every replica executed slots 1 .. last of every group exactly once, in order, and both alike
    for nid in (1, 2):
        nxt = np.ones(G, np.int64)
        for g, first, cnt in ex[nid]:
            assert first == nxt[g]
            nxt[g] += cnt
        trace["next%d" % nid] = nxt.tolist()
        for g in range(G):
            assert all((g, s) in value for s in range(1, int(nxt[g])))
    assert trace["next1"] == trace["next2"]
    cl.close()
    return trace
