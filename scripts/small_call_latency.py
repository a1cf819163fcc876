#!/usr/bin/env python
"""Wall-clock latency of the HOST-pointer entry points on tiny batches (what gpx::PaxosManager and a JNI
host pay per call when a node has next to nothing queued).  Not the judged bench."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gigapaxos_amd import Engine, hri_create, load_hip, S_OK, C_HASVALUE  # noqa: E402
from gigapaxos_amd import wire as W  # noqa: E402


def main():
    out = {}
    for G in (16, 1_000_000):
        e = Engine(load_hip(), 100, G, kmax=3, window=8, max_batch=1 << 16)
        we = W.WireEngine(e)
        ng = min(G, 1024)
        mem = np.tile(np.array([100, 101, 102], np.int32), (ng, 1))
        assert (e.create_groups(np.arange(ng), mem, 3, hri_create(ng, 3, 100)) == S_OK).all()
        names = [b"g%d" % i for i in range(ng)]
        assert (we.bind(names, np.arange(ng)) == S_OK).all()
        g = np.array([3], np.int32)
        res = {}

        def timed(name, fn, reps=200):
            fn()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            res[name] = round((time.perf_counter() - t0) / reps * 1e6, 1)

        slot = [1]

        def round_trip():
            s, bn, bc, med, st = e.propose(g)
            (rb, rc, rm, rf, ast), runs = e.accept(g, bn, bc, s, med)
            for acc in (100, 101):
                d = e.accept_reply(g, rb, rc, s, [acc], rm)
            e.commit(d.gidx, d.bnum, d.bcoord, d.slot, d.median_cp, np.full(d.gidx.shape[0], C_HASVALUE, np.uint8))
            slot[0] += 1

        timed("propose+accept+2x accept_reply+commit (one slot, one group)", round_trip)
        fr = [W.batched_commit(b"g3", 0, 0, 100, 0, [10 ** 6], [101, 102])]
        timed("wire_decode (1 frame)", lambda: we.decode(fr))
        timed("gap_scan (1 group)", lambda: W.gap_scan(we, g, 1))
        timed("poke_scan (1 group)", lambda: e.poke_scan(g))
        out["groups=%d" % G] = res
        e.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
