#!/usr/bin/env python
"""Whole-protocol round on one MI355X (not the judged bench line; numbers go to docs/HISTORY.md §5).

Three engines = the three replicas of every group (BASELINE config #2's shape at config #3's size),
all on one GPU, columns resident in HBM.  One round =
  coordinator: propose(G)                       -> ACCEPT (slot, ballot) per group
  3 acceptors: accept(G) each                   -> replies
  coordinator: accept_reply(3 G votes)          -> G decisions.  The votes are the three acceptors' reply
               columns CONCATENATED - three ascending runs, what a coordinator really receives (each acceptor's
               replies leave gpx_accept_batch grouped by group): the sorted-runs path, no partition
               (gpx_runs.hip.h); --shuffled-replies shuffles them as round 2's bench did (partition pipeline)
  3 replicas:  commit(G) each                   -> in-order execution runs
Per-phase GPU time with torch events; per-kernel split from the engines' hipEvent profile."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gigapaxos_amd import (Engine, hri_create, load_hip, S_OK, C_HASVALUE, ORDERED_PROPOSE,  # noqa: E402
                           ORDERED_ACCEPT, ORDERED_COMMIT, ORDERED_REPLY_RUNS, LAZY_OUTPUTS)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--groups", type=int, default=1_000_000)
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--profile-rounds", type=int, default=4)
    ap.add_argument("--unordered", action="store_true",
                    help="ACCEPT and COMMIT batches in a random record order (what an acceptor sees when the frames "
                         "of many coordinators interleave): the partition path instead of the direct one")
    ap.add_argument("--shuffled-replies", action="store_true",
                    help="the 3 G votes in a random order (round 2's bench): the partition pipeline")
    ap.add_argument("--no-promise", action="store_true",
                    help="do not declare the batches ordered (gpx_engine_set_ordered_batches): the engine then "
                         "also launches the partition path, which returns at once")
    ap.add_argument("--dense-always", action="store_true",
                    help="do not set GPX_LAZY_OUTPUTS: the compaction pass is launched behind every call (it finds "
                         "nothing to do in these rounds); default: lazy, the counts are read once per round anyway")
    args = ap.parse_args()
    G, K = args.groups, 3
    ids = [100, 101, 102]
    dev = torch.device("cuda:0")
    ts = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(ts)
    mem = np.tile(np.array(ids, np.int32), (G, 1))
    eng = {}
    for nid in ids:
        e = Engine(load_hip(), nid, G, kmax=K, window=8, max_batch=3 * G + 1024)
        assert (e.create_groups(np.arange(G, dtype=np.int32), mem, K, hri_create(G, K, 100)) == S_OK).all()
        e.set_stream(ts.cuda_stream)
        if not args.no_promise and not args.unordered:  # every batch here is the previous stage's output: grouped by group
            e.set_ordered_batches(ORDERED_PROPOSE | ORDERED_ACCEPT | ORDERED_COMMIT |
                                  (0 if args.shuffled_replies else ORDERED_REPLY_RUNS) |
                                  (0 if args.dense_always else LAZY_OUTPUTS))
        eng[nid] = e
    i32 = lambda n: torch.empty(n, dtype=torch.int32, device=dev)  # noqa: E731
    u8 = lambda n: torch.empty(n, dtype=torch.uint8, device=dev)  # noqa: E731
    P = lambda t: t.data_ptr()  # noqa: E731
    g_all = torch.arange(G, dtype=torch.int32, device=dev)
    p_slot, p_bnum, p_bcoord, p_med, p_st = i32(G), i32(G), i32(G), i32(G), u8(G)
    rep = {nid: [i32(G), i32(G), i32(G), u8(G), u8(G)] for nid in ids}       # r_bnum r_bcoord r_maxcp r_flags status
    runs = {nid: [i32(G), i32(G), i32(G), torch.zeros(1, dtype=torch.int32, device=dev)] for nid in ids}
    v = [i32(3 * G) for _ in range(6)]
    d = [i32(3 * G) for _ in range(5)] + [u8(3 * G)]
    n_out, v_st = torch.zeros(1, dtype=torch.int32, device=dev), u8(3 * G)
    ckind = torch.full((G,), C_HASVALUE, dtype=torch.uint8, device=dev)
    c_st = u8(G)
    rng = np.random.default_rng(0)
    perms = [torch.from_numpy(rng.permutation(3 * G)).to(dev) for _ in range(2)]
    acc_col = torch.cat([torch.full((G,), nid, dtype=torch.int32, device=dev) for nid in ids])
    gperm = torch.from_numpy(rng.permutation(G)).to(dev) if args.unordered else None
    a_in = [i32(G) for _ in range(5)]   # unordered ACCEPT batch: gidx, bnum, bcoord, slot, median
    c_in = [i32(G) for _ in range(5)]   # unordered COMMIT batch: gidx, bnum, bcoord, slot, median
    t = {"propose": 0.0, "accept_x3": 0.0, "accept_reply": 0.0, "commit_x3": 0.0}
    timed = args.rounds - 1
    for r in range(args.rounds + args.profile_rounds):
        if r == args.rounds:  # per-kernel split in extra rounds: the events cost time, so not in the timed ones
            for e in eng.values():
                e.profile(2)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record()
        eng[100].call_dev("propose_batch", G, P(g_all), 0, P(p_slot), P(p_bnum), P(p_bcoord), P(p_med), P(p_st))
        if gperm is not None:
            for dst, src in zip(a_in, (g_all, p_bnum, p_bcoord, p_slot, p_med)):
                dst.copy_(src[gperm])
        ev[1].record()
        ag, ab, ac_, asl, am = (a_in if gperm is not None else (g_all, p_bnum, p_bcoord, p_slot, p_med))
        for nid in ids:
            rb, rc, rm, rf, st = rep[nid]
            xg, xf, xc, nr = runs[nid]
            eng[nid].call_dev("accept_batch", G, P(ag), P(ab), P(ac_), P(asl), P(am), 0,
                              P(rb), P(rc), P(rm), P(rf), P(st), P(xg), P(xf), P(xc), P(nr))
        ev[2].record()
        # the replies of the three acceptors as the coordinator's vote columns: concatenated, or shuffled
        cat = lambda k: torch.cat([rep[nid][k] for nid in ids])  # noqa: E731
        if args.shuffled_replies or args.unordered:
            pm = perms[r & 1]
            v[0].copy_(ag.repeat(3)[pm]); v[1].copy_(cat(0)[pm]); v[2].copy_(cat(1)[pm])
            v[3].copy_(asl.repeat(3)[pm]); v[4].copy_(acc_col[pm]); v[5].copy_(cat(2)[pm])
        else:
            v[0].copy_(ag.repeat(3)); v[1].copy_(cat(0)); v[2].copy_(cat(1))
            v[3].copy_(asl.repeat(3)); v[4].copy_(acc_col); v[5].copy_(cat(2))
        ev2b = torch.cuda.Event(enable_timing=True)
        ev2b.record()
        eng[100].call_dev("accept_reply_batch", 3 * G, *[P(x) for x in v], *[P(x) for x in d], P(n_out), P(v_st))
        if gperm is not None:
            for dst, src in zip(c_in, (d[0], d[2], d[3], d[1], d[4])):
                dst.copy_(src[:G][gperm])
        ev[3].record()
        cg, cb, cc_, csl, cm = (c_in if gperm is not None else (d[0], d[2], d[3], d[1], d[4]))
        for nid in ids:
            xg, xf, xc, nr = runs[nid]
            eng[nid].call_dev("commit_batch", G, P(cg), P(cb), P(cc_), P(csl), P(cm), P(ckind), P(c_st),
                              P(xg), P(xf), P(xc), P(nr))
        ev[4].record()
        for e in eng.values():
            e.sync()
        torch.cuda.synchronize()
        assert int(n_out) == G, int(n_out)
        for nid in ids:
            assert int(runs[nid][3]) == G and bool((runs[nid][2][:G] == 1).all())
        if 0 < r < args.rounds:
            t["propose"] += ev[0].elapsed_time(ev[1])
            t["accept_x3"] += ev[1].elapsed_time(ev[2])
            t["accept_reply"] += ev2b.elapsed_time(ev[3])
            t["commit_x3"] += ev[3].elapsed_time(ev[4])
    k = timed
    kern = {}
    for nid, e in eng.items():
        for name, (cnt, ms) in e.profile_read().items():
            kern[name] = kern.get(name, 0.0) + ms * 1e3 / max(args.profile_rounds, 1)
    tot = sum(t.values()) / k
    print(json.dumps({"groups": G, "replicas": K, "ordered_batches_promise": not args.no_promise and not args.unordered,
                      "lazy_outputs": not args.dense_always and not args.no_promise and not args.unordered,
                      "acceptor_batches": "unordered" if args.unordered else "grouped by group",
                      "replies": "shuffled" if (args.shuffled_replies or args.unordered) else "three ascending runs",
                      "ms_per_round": round(tot, 4),
                      "phases_ms": {a: round(b / k, 4) for a, b in t.items()},
                      "decided_and_executed_per_s": round(G / tot * 1e3, 1),
                      "kernels_us_per_round": {a: round(b, 1) for a, b in sorted(kern.items())}}))


if __name__ == "__main__":
    main()
