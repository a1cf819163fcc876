#!/usr/bin/env python
"""Writes the coordinator streams RefFixtureDump.java replays (big-endian int32) and, with --collect,
turns its output into tests/golden/ref_<name>.npz.  The streams are gigapaxos_amd.streams.vote_round
rounds (adversarial mix) at a size the JVM finishes in seconds; tests/test_ref_fixtures.py regenerates
the same streams from the seeds stored in the .npz and replays them on the oracle (CPU) and the engine
(GPU)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gigapaxos_amd import streams  # noqa: E402

CASES = {  # name -> (G, K, rounds, config_id)
    "config3_k3": (20000, 3, 6, 3),
    "config4_k5": (12000, 5, 6, 4),
}


def rounds_of(name):
    G, K, R, cfg = CASES[name]
    members = list(range(100, 100 + K))
    out = []
    for r in range(R):
        out.append((np.arange(G, dtype=np.int32), streams.vote_round(G, members, r, 100, config_id=cfg, mix=True)))
    return G, K, 100, members, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--outdir", default="/tmp/ref_fixtures")
    ap.add_argument("--collect", action="store_true", help="read <outdir>/<name>.out and write tests/golden/ref_<name>.npz")
    args = ap.parse_args()
    os.makedirs(args.outdir, exist_ok=True)
    for name in CASES:
        G, K, me, members, rounds = rounds_of(name)
        if not args.collect:
            with open(os.path.join(args.outdir, name + ".in"), "wb") as f:
                np.array([G, K, me, len(rounds)] + members, ">i4").tofile(f)
                for pg, cols in rounds:
                    np.array([pg.shape[0]], ">i4").tofile(f)
                    pg.astype(">i4").tofile(f)
                    np.array([cols[0].shape[0]], ">i4").tofile(f)
                    for c in cols:
                        c.astype(">i4").tofile(f)
            continue
        raw = np.fromfile(os.path.join(args.outdir, name + ".out"), ">i4").astype(np.int32)
        p, props, decs = 0, [], []
        for pg, cols in rounds:
            n = pg.shape[0]
            props.append(raw[p:p + 5 * n].reshape(n, 5))
            p += 5 * n
            nd = int(raw[p])
            p += 1
            decs.append(raw[p:p + 7 * nd].reshape(nd, 7))
            p += 7 * nd
        assert p == raw.shape[0]
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_%s.npz" % name),
                            case=name, **{"prop%d" % i: a for i, a in enumerate(props)},
                            **{"dec%d" % i: a for i, a in enumerate(decs)})
        print("wrote tests/golden/ref_%s.npz: %d rounds, %d decisions" % (name, len(rounds), sum(d.shape[0] for d in decs)))


if __name__ == "__main__":
    main()
