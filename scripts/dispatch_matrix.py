#!/usr/bin/env python
"""Which kernels an accept-reply call of each shape takes, what they cost, and what the OTHER front ends would have cost
for the same call (forced through their switches) - the table tests/test_dispatch_gpu.py's expectations come from.
    python scripts/dispatch_matrix.py [--json OUT]
Paths: default (the dispatcher's choice); GPX_AR_TILES=0 (partition front end: k_hist + k_scatter_ar16); GPX_TRY_RUNS=1
(the runs check first, the dispatcher's choice behind its gate).  Times are hipEvent brackets around each launch (gpx_profile_read), microseconds per call,
median of the timed calls."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gigapaxos_amd import Engine, hri_create, load_hip, streams, S_OK  # noqa: E402

SHAPES = {
    "headline": dict(G=1_000_000, K=3),
    "sorted": dict(G=1_000_000, K=3, shuffled=False),
    "runs": dict(G=1_000_000, K=3, runs=True),
    "mix": dict(G=1_000_000, K=3, mix=True),
    "k5": dict(G=1_000_000, K=5),
    "shard125k_k5": dict(G=125_000, K=5),
    "500k": dict(G=500_000, K=3),
    "odd_first": dict(G=1_000_000, K=3, odd_first=True),
    "out_of_lock_step": dict(G=1_000_000, K=3, lockstep=False),
}
PATHS = {"default": {}, "partition": {"GPX_AR_TILES": "0"}, "runs_hint": {"GPX_TRY_RUNS": "1"}}


def round_cols(shape, r, rng, slot_g):
    G, K = shape["G"], shape["K"]
    members = list(range(100, 100 + K))
    if shape.get("runs"):
        cols = streams.vote_round_runs(G, members, r, 100, config_id=3)
    else:
        cols = streams.vote_round(G, members, r, 100, config_id=3 if K == 3 else 4, shuffled=shape.get("shuffled", True),
                                  mix=shape.get("mix", False))
    cols = [c.copy() for c in cols]
    if not shape.get("lockstep", True):  # every group at its own slot: slot and max_cp from the proposals
        cols[3] = slot_g[cols[0]]
        cols[5] = cols[3] - 1 - rng.integers(0, 300, cols[0].shape[0]).astype(np.int32)
    if shape.get("odd_first"):
        cols[1][0] = 1
        cols[3][0] += 5000
    return [np.ascontiguousarray(c) for c in cols]


def measure(name, shape, env, rounds=6, warm=2):
    """-> ({kernel: us per launch, median over the timed accept-reply calls}, us per call: median of the timed calls)"""
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        G, K = shape["G"], shape["K"]
        members = list(range(100, 100 + K))
        rng = np.random.default_rng(1)
        dev = torch.device("cuda:0")
        e = Engine(load_hip(), 100, G, kmax=K, window=8, max_batch=G * K + G * K // 25 + 4096)
        mem = np.tile(np.array(members, np.int32), (G, 1))
        rows = hri_create(G, K, 100)
        if not shape.get("lockstep", True):
            base = rng.integers(1, 2_000_000, G).astype(np.int32)
            rows["acc_slot"], rows["acc_gc_slot"], rows["next_proposal_slot"] = base, base - 2, base
            rows["node_slots"][:, :K] = (base - 2)[:, None]
        assert (e.create_groups(np.arange(G, dtype=np.int32), mem, K, rows) == S_OK).all()
        g = np.arange(G, dtype=np.int32)
        P = lambda t: t.data_ptr()  # noqa: E731
        times, percall = [], []
        for r in range(rounds):
            slot_g = e.propose(g)[0]
            cols = round_cols(shape, r, rng, slot_g)
            n = cols[0].shape[0]
            dc = [torch.from_numpy(c).to(dev) for c in cols]
            d = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(5)] + [torch.zeros(n, dtype=torch.uint8, device=dev)]
            no, st = torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            e.profile(2)
            e.call_dev("accept_reply_batch", n, *[P(c) for c in dc], *[P(t) for t in d], P(no), P(st))
            e.sync()
            prof = e.profile_read()
            e.profile(0)
            if r >= warm:
                times.append(sum(ms for _, ms in prof.values()) * 1e3)
                percall.append({k: ms * 1e3 / max(nl, 1) for k, (nl, ms) in prof.items()})  # us per launch
        e.close()
        # per kernel the MEDIAN over the timed calls (one call's bracket can be off by half its length on a small shape:
        # the 125,000-group shard's scatter once read 20 us for 13 and "dominated" the call)
        kernels = {k: round(float(np.median([c[k] for c in percall if k in c])), 1) for k in percall[-1]}
        return kernels, float(np.median(times))
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--shapes", default=",".join(SHAPES))
    a = ap.parse_args()
    table = {}
    for name in a.shapes.split(","):
        shape = SHAPES[name]
        row = {}
        for pname, env in PATHS.items():
            k, us = measure(name, shape, env)
            row[pname] = {"kernels": k, "us": round(us, 1), "dominant": max(k, key=k.get)}
            print(f"{name:18s} {pname:10s} {us:8.1f} us   {' '.join('%s=%.1f' % kv for kv in sorted(k.items()))}", flush=True)
        table[name] = row
    if a.json:
        json.dump(table, open(a.json, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
