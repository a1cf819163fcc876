"""Differential driver: applies one seeded op sequence to two engines behind the same C-ABI
(the HIP engine and the CPU oracle) and compares every output column and the full per-group
state dump bit for bit."""
import numpy as np

from gigapaxos_amd import Engine, hri_create, hri_initial, make_hri, streams, S_OK, S_WINDOW


def wrap32(x):
    """Java int wraparound of an int64 value (array or scalar)."""
    return (np.asarray(x, np.int64) & 0xFFFFFFFF).astype(np.uint32).view(np.int32)


def make_pair(lib_a, lib_b, my_id, G, kmax, window, max_batch=1 << 16, flags=1):
    return (Engine(lib_a, my_id, G, kmax=kmax, window=window, max_batch=max_batch, flags=flags),
            Engine(lib_b, my_id, G, kmax=kmax, window=window, max_batch=max_batch, flags=flags))


def assert_same_state(ea, eb, groups):
    for g in groups:
        da, db = ea.dump(int(g)), eb.dump(int(g))
        assert da.tolist() == db.tolist(), f"group {g} state differs:\n{da.tolist()}\n{db.tolist()}"


def create_mixed_groups(ea, eb, G, kmax, node_ids, rng, slot_base=1, my_id=None):
    """Groups of varying size k <= kmax, members drawn (sorted) from node_ids; half created with
    createHRI rows, half with regular-creation rows; coordinator = my_id for ~2/3 of them."""
    my_id = ea.my_id if my_id is None else my_id
    gidx = np.arange(G, dtype=np.int32)
    members = np.zeros((G, kmax), np.int32)
    ks = np.zeros(G, np.uint8)
    rows = make_hri(G)
    for g in range(G):
        k = int(rng.integers(1, kmax + 1))
        others = [n for n in node_ids if n != my_id]
        mem = sorted([my_id] + list(rng.choice(others, size=k - 1, replace=False))) if k > 1 else [my_id]
        ks[g] = k
        members[g, :k] = mem
        coord = my_id if rng.random() < 0.67 else int(rng.choice(mem))
        r = (hri_create if g % 2 == 0 else hri_initial)(1, k, coord)
        r["acc_slot"] = wrap32(int(r["acc_slot"][0]) - 1 + slot_base)
        r["acc_gc_slot"] = wrap32(int(r["acc_gc_slot"][0]) - 1 + slot_base)
        r["next_proposal_slot"] = wrap32(int(r["next_proposal_slot"][0]) - 1 + slot_base)
        if g % 2 == 0:
            r["node_slots"][0, :k] = wrap32(np.full(k, slot_base - 1))
        rows[g] = r[0]
    sa = ea.create_groups(gidx, members, ks, rows)
    sb = eb.create_groups(gidx, members, ks, rows)
    assert sa.tolist() == sb.tolist() and (sa == S_OK).all()
    return members, ks


def fuzz(ea, eb, G, node_ids, rng, steps, batch, slot_base=1, span=40, my_id=None, p_stop=0.01,
         ordered=False):
    """Random interleaving of propose / accept / accept_reply / commit batches with colliding
    slots, duplicate votes, stale and higher ballots, non-member acceptors, unknown groups."""
    my_id = ea.my_id if my_id is None else my_id
    nodes = np.array(list(node_ids) + [my_id - 7], np.int32)  # last = never a member

    def gids(n):
        g = rng.integers(0, G, n).astype(np.int32)
        if ordered:
            # grouped by group, groups ascending (what the previous pipeline stage emits): the engine
            # applies such ACCEPT / COMMIT batches without partitioning them; one batch in eight keeps
            # an out-of-range index, which sends it down the partition path instead
            g.sort()
            if rng.random() < 0.125:
                g[rng.integers(0, n)] = rng.choice([-1, G, G + 5])
            return g
        bad = rng.random(n) < 0.01
        g[bad] = rng.choice([-1, G, G + 5], size=int(bad.sum()))
        return g

    def slots(n):
        return wrap32(slot_base + rng.integers(-2, span, n))

    def ballots(n):
        bnum = rng.choice([0, 0, 0, 0, 1, 2], size=n).astype(np.int32)
        bcoord = rng.choice(nodes[:-1], size=n).astype(np.int32)
        pref = rng.random(n) < 0.7
        bcoord[pref] = my_id
        bnum[pref & (rng.random(n) < 0.9)] = 0
        return bnum, bcoord

    for step in range(steps):
        op = rng.integers(0, 5)
        n = int(rng.integers(1, batch + 1))
        g = gids(n)
        if op == 0:
            # keep every group's proposal frontier inside the slot span the votes can reach, so
            # the engine's fixed window never fills (the oracle's maps are unbounded): at most one
            # proposal per group per batch, only for groups whose next slot is still in range
            rows, _ = ea.snapshot(np.arange(G))
            room = (rows["next_proposal_slot"].astype(np.int64) - slot_base) < span - 2
            ok = (g < 0) | (g >= G)
            inr = ~ok
            ok[inr] = room[g[inr]] | (rows["has_coord"][g[inr]] == 0)
            _, first = np.unique(g, return_index=True)
            uniq = np.zeros(n, bool)
            uniq[first] = True
            g = g[ok & uniq]
            n = g.shape[0]
            if n == 0:
                continue
            stop = (rng.random(n) < p_stop).astype(np.uint8) if rng.random() < 0.5 else None
            ra, rb = ea.propose(g, stop), eb.propose(g, stop)
            for x, y, nm in zip(ra, rb, ("slot", "bnum", "bcoord", "median", "status")):
                assert x.tolist() == y.tolist(), f"step {step} propose {nm}"
        elif op == 1:
            bnum, bcoord = ballots(n)
            sl = slots(n)
            med = wrap32(slot_base + rng.integers(-3, span, n))
            fl = (rng.random(n) < p_stop).astype(np.uint8)
            (ra, xa), (rb, xb) = ea.accept(g, bnum, bcoord, sl, med, fl), eb.accept(g, bnum, bcoord, sl, med, fl)
            for x, y, nm in zip(ra, rb, ("r_bnum", "r_bcoord", "r_maxcp", "r_flags", "status")):
                assert x.tolist() == y.tolist(), f"step {step} accept {nm}"
            assert xa.as_tuple_array().tolist() == xb.as_tuple_array().tolist(), f"step {step} accept runs"
        elif op == 2:
            bnum, bcoord = ballots(n)
            sl = slots(n)
            acc = rng.choice(nodes, size=n).astype(np.int32)
            mcp = wrap32(slot_base + rng.integers(-3, span, n))
            da, db = ea.accept_reply(g, bnum, bcoord, sl, acc, mcp), eb.accept_reply(g, bnum, bcoord, sl, acc, mcp)
            assert da.as_tuple_array().tolist() == db.as_tuple_array().tolist(), f"step {step} decisions"
            assert da.status.tolist() == db.status.tolist(), f"step {step} ar status"
        elif op == 4:
            # PREPAREs (view change, acceptor side): ballots around the current ones, so that acks,
            # NACKs and ballot upgrades all occur; firstUndecidedSlot around the live slots
            bnum, bcoord = ballots(n)
            first = slots(n)
            (ra, pa), (rb, pb) = ea.prepare(g, bnum, bcoord, first), eb.prepare(g, bnum, bcoord, first)
            for x, y, nm in zip(ra, rb, ("r_bnum", "r_bcoord", "r_gc", "r_flags", "status")):
                assert x.tolist() == y.tolist(), f"step {step} prepare {nm}"
            assert pa == pb, f"step {step} prepare pvalues"
        else:
            bnum, bcoord = ballots(n)
            sl = slots(n)
            med = wrap32(slot_base + rng.integers(-3, span, n))
            kind = rng.choice([0, 0, 0, 1, 3], size=n).astype(np.uint8)
            (sa, xa), (sb, xb) = ea.commit(g, bnum, bcoord, sl, med, kind), eb.commit(g, bnum, bcoord, sl, med, kind)
            assert sa.tolist() == sb.tolist(), f"step {step} commit status"
            assert xa.as_tuple_array().tolist() == xb.as_tuple_array().tolist(), f"step {step} commit runs"
    assert_same_state(ea, eb, range(G))
    ca, cb = ea.counters(), eb.counters()
    assert ca == cb, (ca, cb)


def churn_run(engines, G_live, cap, R, k, seed, churn_frac=0.01):
    """BASELINE config #5 driver (reconfiguration churn): every round proposes + votes for all live
    groups, then retires churn_frac of them (PaxosManager.kill) and creates the same number on
    fresh gidx rows (free list); the next round's stream still carries votes for the groups just
    retired, which must be dropped with GPX_S_NOGROUP (PaxosManager.java:1162-1194).  Applies the
    identical op sequence to every engine in `engines` and returns, per engine, the list of
    (decisions, vote status, propose status, retire rows, create status) per round."""
    from gigapaxos_amd import RETIRE_KILL
    members = list(range(100, 100 + k))
    rng = np.random.default_rng(seed)
    from collections import deque
    live = np.arange(G_live, dtype=np.int32)
    free = deque(range(G_live, cap))  # FIFO: a retired row is reused a few rounds later
    outs = [[] for _ in engines]
    for e in engines:
        mem = np.tile(np.array(members, np.int32), (G_live, 1))
        assert (e.create_groups(live, mem, k, hri_create(G_live, k, 100)) == S_OK).all()
    ghost = np.zeros(0, np.int32)  # retired last round: still addressed by late votes
    slot_of = np.ones(cap, np.int64)  # next slot each gidx row will propose (fresh rows restart at 1)
    for r in range(R):
        rng_r = np.random.default_rng(seed * 1000 + r)
        pg = rng_r.permutation(np.concatenate([live, ghost])).astype(np.int32)
        cols = streams.vote_round(0, members, 0, 100, config_id=5, groups=np.concatenate([live, ghost]))
        cols = list(cols)
        cols[3] = slot_of[cols[0]].astype(np.int32)          # slot = the row's outstanding slot
        cols[5] = (slot_of[cols[0]] - 1).astype(np.int32)    # max_cp = slot - 1
        order = rng_r.permutation(cols[0].shape[0])
        cols = [np.ascontiguousarray(c[order]) for c in cols]
        n_ret = max(1, int(live.shape[0] * churn_frac))
        victims = rng_r.choice(live, size=n_ret, replace=False).astype(np.int32)
        fresh = np.array([free.popleft() for _ in range(n_ret)], np.int32)
        for e, out in zip(engines, outs):
            pr = e.propose(pg)
            d = e.accept_reply(*cols)
            rows, st_ret = e.retire_groups(victims, RETIRE_KILL)
            mem = np.tile(np.array(members, np.int32), (n_ret, 1))
            st_new = e.create_groups(fresh, mem, k, hri_create(n_ret, k, 100))
            out.append((d.as_tuple_array(), d.status, np.stack(pr[:4], 1), pr[4], rows.tobytes(),
                        st_ret, st_new))
        slot_of[live] += 1
        keep = np.ones(cap, bool)
        keep[victims] = False
        live = np.concatenate([live[keep[live]], fresh])
        slot_of[fresh] = 1
        free.extend(int(v) for v in victims)
        ghost = victims
    return outs, live


def steady_state_run(lib_a, lib_b, kmax, seed, node_ids, G=6000, rounds=6):
    """Accept-reply batches in the coordinator's steady state (every group's votes answer one outstanding
    slot at the current ballot), with the variety that must not matter: 1..4 (1..10) votes per group,
    duplicates, non-members, older outstanding slots, group sizes kmax-2..kmax, and from round 4 on a few
    votes of another ballot."""
    rng = np.random.default_rng(seed)
    nodes = list(node_ids)[:kmax]
    eh, eo = make_pair(lib_a, lib_b, 100, G, kmax, 8, max_batch=1 << 16)
    ks = rng.integers(max(1, kmax - 2), kmax + 1, G).astype(np.uint8)          # group sizes kmax-2 .. kmax
    mem = np.zeros((G, kmax), np.int32)
    for k in np.unique(ks):
        mem[ks == k, :k] = np.array(nodes[:k], np.int32)                         # node 100 first: it coordinates
    for e in (eh, eo):
        st = e.create_groups(np.arange(G), mem, ks, hri_create(G, kmax, 100))
        assert (st == S_OK).all()
    g = np.arange(G, dtype=np.int32)
    nonmember = 7777
    decided = 0
    for r in range(rounds):
        for _ in range(2 if r % 2 else 1):                                       # one or two slots outstanding
            for x, y in zip(eh.propose(g), eo.propose(g)):
                assert (x == y).all()
        rows, _ = eo.snapshot(g)
        newest = rows["next_proposal_slot"].astype(np.int32) - 1
        target = np.where((rng.random(G) < 0.3) & (r % 2 == 1), newest - 1, newest).astype(np.int32)
        c = rng.integers(1, 11 if kmax >= 5 else 5, G)                           # 1..4 (1..10) votes per group: past the path's limit of 8 too
        gi = np.repeat(g, c)
        n = gi.shape[0]
        pick = rng.integers(0, kmax + 1, n)                                      # column kmax = a node that is no member
        acc = np.where(pick < ks[gi], mem[gi, np.minimum(pick, kmax - 1)], nonmember).astype(np.int32)
        cols = [gi, np.zeros(n, np.int32), np.full(n, 100, np.int32), target[gi],
                acc, rng.integers(-1, 9, n).astype(np.int32)]
        if r >= 4:                                                               # a few votes of a higher / lower ballot
            odd = rng.integers(0, n, 20)
            cols[1][odd] = rng.choice([0, 1], 20)
            cols[2][odd] = rng.choice([99, 101], 20)
        order = rng.permutation(n)
        cols = [np.ascontiguousarray(x[order]) for x in cols]
        dh, do = eh.accept_reply(*cols), eo.accept_reply(*cols)
        assert (dh.as_tuple_array() == do.as_tuple_array()).all(), f"round {r}"
        assert (dh.status == do.status).all(), f"round {r} status"
        decided += dh.gidx.shape[0]
    assert decided > (G if kmax <= 5 else 100)
    sh, so = eh.snapshot(g)[0], eo.snapshot(g)[0]
    assert sh.tobytes() == so.tobytes()
    assert eh.counters() == eo.counters()
    eh.close()
    eo.close()
