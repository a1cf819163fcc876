#!/usr/bin/env python
"""A bare loop of the headline step (propose + shuffled accept replies on device columns) for the counter passes of
scripts/gpu_visit.sh (pmc:ar_loop.py[:ARGS]) - no CPU leg, no end-to-end leg, nothing else on the device.
    python scripts/ar_loop.py [--groups G] [--k K] [--rounds R] [--mix] [--sorted]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gigapaxos_amd import Engine, hri_create, load_hip, streams, S_OK, ORDERED_PROPOSE  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--groups", type=int, default=1_000_000)
    ap.add_argument("--k", type=int, default=3)
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--mix", action="store_true")
    ap.add_argument("--sorted", action="store_true")
    ap.add_argument("--profile", action="store_true", help="print the kernels' microseconds per launch (hipEvent brackets) of the last rounds")
    a = ap.parse_args()
    G, K = a.groups, a.k
    members = list(range(100, 100 + K))
    dev = torch.device("cuda:0")
    P = lambda t: t.data_ptr()  # noqa: E731
    e = Engine(load_hip(), 100, G, kmax=K, window=8, max_batch=G * K + G * K // 25 + 4096)
    mem = np.tile(np.array(members, np.int32), (G, 1))
    assert (e.create_groups(np.arange(G, dtype=np.int32), mem, K, hri_create(G, K, 100)) == S_OK).all()
    e.set_ordered_batches(ORDERED_PROPOSE)
    ts = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(ts)
    e.set_stream(ts.cuda_stream)
    rounds = [[torch.from_numpy(c).to(dev) for c in streams.vote_round_survey(G, members, r, 100, config_id=3 if K == 3 else 4,
                                                                              shuffled=not a.sorted, mix=a.mix)] for r in range(a.rounds)]
    n = int(rounds[0][0].shape[0])
    g = torch.arange(G, dtype=torch.int32, device=dev)
    p = [torch.empty(G, dtype=torch.int32, device=dev) for _ in range(4)] + [torch.empty(G, dtype=torch.uint8, device=dev)]
    d = [torch.empty(n + 64, dtype=torch.int32, device=dev) for _ in range(5)] + [torch.empty(n + 64, dtype=torch.uint8, device=dev)]
    no, st = torch.zeros(1, dtype=torch.int32, device=dev), torch.empty(n + 64, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for r in range(a.rounds):
        c = rounds[r]
        if a.profile and r == 4:
            e.sync()
            e.profile(2)
        e.call_dev("propose_batch", G, P(g), 0, *[P(t) for t in p])
        e.call_dev("accept_reply_batch", int(c[0].shape[0]), *[P(t) for t in c], *[P(t) for t in d], P(no), P(st))
    e.sync()
    torch.cuda.synchronize()
    print("decisions of the last round:", int(no))
    if a.profile:
        print({k: round(ms * 1e3 / max(nl, 1), 1) for k, (nl, ms) in sorted(e.profile_read().items())})
    e.close()


if __name__ == "__main__":
    main()
