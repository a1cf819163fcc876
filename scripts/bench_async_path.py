#!/usr/bin/env python
"""The asynchronous host-pointer calls at the bench size: where does a step's time go?  One step =
gpx_propose_batch_async(1 M) + gpx_accept_reply_batch_async(3 M shuffled votes) from registered host memory,
DEPTH steps in flight (1 = submit and wait at once: same streams, no overlap), with or without ballot columns.
Host time inside submit() and wait() is reported separately (a runtime that copies synchronously shows up as
submit time).  Engine experiments by environment: GPX_ASYNC_IN=engine, GPX_ASYNC_FILL=memset."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gigapaxos_amd import Engine, hri_create, load_hip, streams, S_OK  # noqa: E402


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def run(depth, common, G, K, steps, sync=False):
    lib = load_hip()
    members = list(range(100, 100 + K))
    nv = G * K
    e = Engine(lib, 100, G, kmax=K, window=8, max_batch=nv + 1024)
    mem = np.tile(np.array(members, np.int32), (G, 1))
    assert (e.create_groups(np.arange(G, dtype=np.int32), mem, K, hri_create(G, K, 100)) == S_OK).all()
    hg = np.arange(G, dtype=np.int32)
    rounds = [[np.ascontiguousarray(c) for c in streams.vote_round(G, members, r, 100)] for r in range(steps + 1)]
    ring = []
    for _ in range(max(depth, 1)):
        o = [np.zeros(G, np.int32) for _ in range(4)] + [np.zeros(G, np.uint8)]
        d = [np.zeros(nv, np.int32) for _ in range(5)] + [np.zeros(nv, np.uint8)]
        ring.append((o, d, np.zeros(1, np.int32), np.zeros(nv, np.uint8)))
    pinned = [hg] + [c for rd in rounds for c in rd] + [x for o, d, no, st in ring for x in o + d + [no, st]]
    e.host_register(*pinned)
    fn = lib.fn
    t_sub = t_wait = 0.0

    def submit(r):
        o, d, no, st = ring[r % len(ring)]
        c = rounds[r]
        if sync:
            rc = fn["propose_batch"](e.h, G, _p(hg), None, *[_p(x) for x in o])
            rc |= fn["accept_reply_batch"](e.h, nv, *[_p(x) for x in c], *[_p(x) for x in d], _p(no), _p(st))
            assert rc == 0
            return None
        tp, ta = C.c_uint64(0), C.c_uint64(0)
        rc = fn["propose_batch_async"](e.h, G, _p(hg), None, *[_p(x) for x in o], C.byref(tp))
        rc |= fn["accept_reply_batch_async"](e.h, nv, _p(c[0]), None if common else _p(c[1]), None if common else _p(c[2]),
                                             0, 100, _p(c[3]), _p(c[4]), _p(c[5]), *[_p(x) for x in d], _p(no), _p(st),
                                             C.byref(ta))
        assert rc == 0, rc
        return tp, ta

    def wait(t):
        if t is not None:
            assert fn["engine_wait"](e.h, t[0]) == 0 and fn["engine_wait"](e.h, t[1]) == 0

    # warm: every set of device columns the timed loop will use is allocated now (first use allocates); the
    # repeated round only brings late votes, and leaves one more slot outstanding per extra warm step
    warm = [submit(0) for _ in range(1 if sync else max(depth, 1))]
    for t in warm:
        wait(t)
    t0 = time.perf_counter()
    pend = []
    for r in range(1, steps + 1):
        a = time.perf_counter()
        pend.append(submit(r))
        b = time.perf_counter()
        t_sub += b - a
        if len(pend) >= depth:
            wait(pend.pop(0))
            t_wait += time.perf_counter() - b
    a = time.perf_counter()
    for t in pend:
        wait(t)
    t_wait += time.perf_counter() - a
    el = (time.perf_counter() - t0) / steps
    assert int(ring[steps % len(ring)][2][0]) == G
    e.host_unregister(*pinned)
    e.close()
    b_in = G * 4 + nv * (16 if (common and not sync) else 24)
    b_out = G * 17 + nv + G * 21 + 4
    return {"ms_per_step": round(el * 1e3, 3), "submit_ms": round(t_sub / steps * 1e3, 3), "wait_ms": round(t_wait / steps * 1e3, 3),
            "in_GBps": round(b_in / el / 1e9, 1), "out_GBps": round(b_out / el / 1e9, 1)}


def main():
    sys.path.insert(0, ROOT)
    import torch
    torch.zeros(1, device="cuda:0")
    from bench import pin_to_gpu_numa_node
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-pin", action="store_true", help="do not move the process next to the GPU")
    ap.add_argument("--groups", type=int, default=1_000_000)
    ap.add_argument("--k", type=int, default=3)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--repeats", type=int, default=2)
    ap.add_argument("--common-first", action="store_true")
    a = ap.parse_args()
    pinned = None if a.no_pin else pin_to_gpu_numa_node(0)
    out = {"env": {k: os.environ.get(k) for k in ("GPX_ASYNC_IN", "GPX_ASYNC_FILL", "GPX_ASYNC_DIRECT", "GPX_ASYNC_COPYIN")},
           "pinned_to_gpu_numa_node": pinned is not None}
    out["sync"] = run(1, False, a.groups, a.k, a.steps, sync=True)
    for rep in range(a.repeats):  # repeats: the first configuration to touch a fresh set of device columns pays for it
        for depth in (1, 2):
            for common in ((True, False) if a.common_first else (False, True)):
                out["async depth %d%s #%d" % (depth, " common ballot" if common else "", rep)] = \
                    run(depth, common, a.groups, a.k, a.steps)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
