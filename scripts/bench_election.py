#!/usr/bin/env python
"""Failover burst on one MI355X (not the judged bench line; numbers go to docs/HISTORY.md 5).

The node that coordinated every group dies.  One surviving node (one engine, G groups, 3 replicas):
  gpx_election_scan(all groups)            who must run
  gpx_election_begin(G)                    new coordinator state, ballot (1, me)
  gpx_propose_batch_h(0.3 G)               client requests while not active: pre-active proposals
  gpx_prepare_reply_batch(2 G replies)     two PREPARE replies per group, shuffled, half of them with
                                           an accepted pvalue -> G view changes, ~1.8 ACCEPTs per group
Host-pointer calls: the wall time includes the PCIe copies of the columns; the per-kernel split
comes from the engine's hipEvent profile."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gigapaxos_amd import Engine, load_hip, make_hri, S_OK  # noqa: E402
from gigapaxos_amd import wire as W  # noqa: E402


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--groups", type=int, default=1_000_000)
    args = ap.parse_args()
    G, K, Wn = args.groups, 3, 8
    torch.cuda.init()  # before the engine's own HIP calls: one runtime, torch's
    lib = load_hip()
    e = Engine(lib, 1, G, kmax=K, window=Wn, max_batch=2 * G + 1024)
    rows = make_hri(G)
    rows["acc_slot"] = 5
    rows["acc_gc_slot"] = 4
    rows["next_proposal_slot"] = -1
    mem = np.tile(np.array([0, 1, 2], np.int32), (G, 1))
    allg = np.arange(G, dtype=np.int32)
    assert (e.create_groups(allg, mem, K, rows) == S_OK).all()
    rng = np.random.default_rng(0)
    we = W.WireEngine(e)
    # the burst three times over (ballots 1, 2, 3): the first pass pays one-off costs (scratch memory
    # for the view-change kernel, lazily allocated tables); the last pass is reported
    for it in range(3):
        e.profile(2)
        t = {}
        t0 = time.perf_counter()
        run, pb, pf, st = W.election_scan(we, None, down_nodes=[0], force=it > 0)
        t["election_scan"] = time.perf_counter() - t0
        assert run.all()
        bn = np.full(G, it + 1, np.int32)
        t0 = time.perf_counter()
        es = e.election_begin(allg, bn)
        t["election_begin"] = time.perf_counter() - t0
        assert (es == 0).all()
        pre = allg[rng.random(G) < 0.3]
        t0 = time.perf_counter()
        r = e.propose(pre, handle=np.arange(1, pre.size + 1, dtype=np.int64))
        t["propose_preactive"] = time.perf_counter() - t0
        assert (r[4] == 8).all()
        # two replies per group (acceptors 1 and 2), shuffled; half carry one accepted pvalue at slot 5 or 6
        n = 2 * G
        perm = rng.permutation(n)
        gi = np.concatenate([allg, allg])[perm].astype(np.int32)
        acc = np.concatenate([np.full(G, 1, np.int32), np.full(G, 2, np.int32)])[perm]
        has = rng.random(n) < 0.5
        off = np.zeros(n + 1, np.int32)
        off[1:] = np.cumsum(has)
        m = int(off[n])
        ps = (5 + rng.integers(0, 2, m)).astype(np.int32)
        pbn, pbc = np.zeros(m, np.int32), np.zeros(m, np.int32)
        ph = (10 ** 9 + np.arange(m)).astype(np.int64)
        pfl = np.zeros(m, np.uint8)
        rb, rc = bn.repeat(2), np.full(n, 1, np.int32)
        first = np.full(n, 5, np.int32)
        vk, stt = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        ec, em = np.zeros(n, np.int32), np.zeros(n, np.int32)
        es_, ek, ef = np.zeros(n * Wn, np.int32), np.zeros(n * Wn, np.uint8), np.zeros(n * Wn, np.uint8)
        eh = np.zeros(n * Wn, np.int64)
        t0 = time.perf_counter()
        lib.check(lib.fn["prepare_reply_batch"](e.h, n, _p(gi), _p(acc), _p(rb), _p(rc), _p(first), _p(off),
                                                _p(ps), _p(pbn), _p(pbc), _p(ph), _p(pfl), _p(vk), _p(ec), _p(em),
                                                _p(es_), _p(ek), _p(eh), _p(ef), _p(stt)), "prepare_reply_batch")
        t["prepare_reply"] = time.perf_counter() - t0
        assert int((vk == 2).sum()) == G and int((vk == 1).sum()) == G and not stt.any()
    # the same burst with every column resident in HBM (the *_dev twins), timed with events on the
    # engine's stream: ballots 4 and 5, the second one reported
    dev = torch.device("cuda:0")
    ts = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(ts)
    e.set_stream(ts.cuda_stream)
    T = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    VP = lambda t_: C.c_void_p(t_.data_ptr())  # noqa: E731
    d_allg, d_pre = T(allg), T(pre)
    d_h = T(np.arange(1, pre.size + 1, dtype=np.int64))
    d_gi, d_acc, d_rc, d_first, d_off = T(gi), T(acc), T(rc), T(first), T(off)
    d_ps, d_pbn, d_pbc, d_ph, d_pfl = T(ps), T(pbn), T(pbc), T(ph), T(pfl)
    o_es = torch.zeros(G, dtype=torch.uint8, device=dev)
    o_p = [torch.zeros(pre.size, dtype=torch.int32, device=dev) for _ in range(4)]
    o_pst = torch.zeros(pre.size, dtype=torch.uint8, device=dev)
    o_vk, o_st = (torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(2))
    o_ec, o_em = (torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(2))
    o_es_ = torch.zeros(n * Wn, dtype=torch.int32, device=dev)
    o_ek, o_ef = (torch.zeros(n * Wn, dtype=torch.uint8, device=dev) for _ in range(2))
    o_eh = torch.zeros(n * Wn, dtype=torch.int64, device=dev)
    resident = {}
    for it in (3, 4):
        d_bn = torch.full((G,), it + 1, dtype=torch.int32, device=dev)
        d_rb = torch.full((n,), it + 1, dtype=torch.int32, device=dev)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        lib.check(lib.fn["election_begin_dev"](e.h, G, VP(d_allg), VP(d_bn), VP(o_es)), "election_begin_dev")
        ev[1].record()
        lib.check(lib.fn["propose_batch_h_dev"](e.h, pre.size, VP(d_pre), None, VP(d_h), VP(o_p[0]), VP(o_p[1]),
                                                VP(o_p[2]), VP(o_p[3]), VP(o_pst)), "propose_batch_h_dev")
        ev[2].record()
        lib.check(lib.fn["prepare_reply_batch_dev"](e.h, n, VP(d_gi), VP(d_acc), VP(d_rb), VP(d_rc), VP(d_first),
                                                    VP(d_off), m, VP(d_ps), VP(d_pbn), VP(d_pbc), VP(d_ph),
                                                    VP(d_pfl), VP(o_vk), VP(o_ec), VP(o_em), VP(o_es_), VP(o_ek),
                                                    VP(o_eh), VP(o_ef), VP(o_st)), "prepare_reply_batch_dev")
        ev[3].record()
        torch.cuda.synchronize()
        assert int((o_vk == 2).sum()) == G and int((o_es == 0).sum()) == G and int((o_pst == 8).sum()) == pre.size
        resident = {"election_begin": ev[0].elapsed_time(ev[1]), "propose_preactive": ev[1].elapsed_time(ev[2]),
                    "prepare_reply": ev[2].elapsed_time(ev[3]), "total": ev[0].elapsed_time(ev[3])}
    prof = e.profile_read()
    out = {
        "workload": f"{G} groups x 3 replicas, every group fails over at once; {n} PREPARE replies, {m} pvalues",
        "elections": G,
        "accepts_spawned": int(ec.sum()),
        "host_call_ms": {k: round(v * 1e3, 3) for k, v in t.items()},
        "resident_gpu_ms": {k: round(v, 4) for k, v in resident.items()},
        "view_changes_per_sec_resident": round(G / (resident["total"] * 1e-3)),
        "kernel_ms_total_3_bursts": {k: round(v[1], 4) for k, v in sorted(prof.items())},
    }
    print(json.dumps(out))
    e.close()


if __name__ == "__main__":
    main()
