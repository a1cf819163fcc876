#!/usr/bin/env python
"""Batch-size sweep of the accept-reply call ("batch sizes swept 2^16 .. 2^24").

1 M groups x 3 replicas on one engine; six rounds are proposed (six slots outstanding per group, the
window holds eight), their 18 M votes - each round shuffled on its own, rounds back to back - are fed
to gpx_accept_reply_batch_dev in chunks of B votes, columns resident in HBM.  Reports votes/s over
the whole stream for every B and checks that every group decided every round."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gigapaxos_amd import Engine, hri_create, load_hip, S_OK  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--groups", type=int, default=1_000_000)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--min-log2", type=int, default=16)
    ap.add_argument("--max-log2", type=int, default=24)
    args = ap.parse_args()
    G, K, R = args.groups, 3, args.rounds
    ids = [100, 101, 102]
    dev = torch.device("cuda:0")
    ts = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(ts)
    P = lambda t_: t_.data_ptr()  # noqa: E731
    g_all = torch.arange(G, dtype=torch.int32, device=dev)
    N = K * G * R
    # the stream: round r = slot r + 1 of every group, one vote per acceptor, shuffled inside the round
    gen = torch.Generator(device=dev)
    gen.manual_seed(12345)
    cols = [torch.empty(N, dtype=torch.int32, device=dev) for _ in range(6)]  # gidx bnum bcoord slot acceptor max_cp
    acc_col = torch.cat([torch.full((G,), nid, dtype=torch.int32, device=dev) for nid in ids])
    for r in range(R):
        pm = torch.randperm(K * G, device=dev, generator=gen)
        sl = slice(r * K * G, (r + 1) * K * G)
        cols[0][sl] = g_all.repeat(K)[pm]
        cols[1][sl] = 0
        cols[2][sl] = 100
        cols[3][sl] = r + 1
        cols[4][sl] = acc_col[pm]
        cols[5][sl] = r
    out = {}
    mem = np.tile(np.array(ids, np.int32), (G, 1))
    d = [torch.empty(N, dtype=torch.int32, device=dev) for _ in range(5)] + [torch.empty(N, dtype=torch.uint8, device=dev)]
    n_out = torch.zeros(1, dtype=torch.int32, device=dev)
    v_st = torch.empty(N, dtype=torch.uint8, device=dev)
    p = [torch.empty(G, dtype=torch.int32, device=dev) for _ in range(4)] + [torch.empty(G, dtype=torch.uint8, device=dev)]
    for lb in range(args.min_log2, args.max_log2 + 1):
        B = 1 << lb
        e = Engine(load_hip(), 100, G, kmax=K, window=8, max_batch=max(B, G) + 1024)
        assert (e.create_groups(np.arange(G, dtype=np.int32), mem, K, hri_create(G, K, 100)) == S_OK).all()
        e.set_stream(ts.cuda_stream)
        for r in range(R):
            e.call_dev("propose_batch", G, P(g_all), 0, *[P(x) for x in p])
        e.sync()
        torch.cuda.synchronize()
        assert bool((p[4] == 0).all()) and bool((p[0] == R).all())
        total = 0
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        counts = []
        ev0.record()
        for o in range(0, N, B):
            nn = min(B, N - o)
            e.call_dev("accept_reply_batch", nn, *[P(c[o:]) for c in cols], *[P(x) for x in d], P(n_out), P(v_st[o:]))
            counts.append(n_out.clone())  # stream-ordered copy, no host sync inside the timed region
        ev1.record()
        e.sync()
        torch.cuda.synchronize()
        total = int(torch.stack(counts).sum())
        assert total == G * R, (B, total)
        ms = ev0.elapsed_time(ev1)
        out[f"2^{lb}"] = {"calls": len(counts), "ms_total": round(ms, 3), "us_per_call": round(ms * 1e3 / len(counts), 1),
                          "votes_per_sec": round(N / ms * 1e3), "decisions_per_sec": round(G * R / ms * 1e3)}
        e.close()
    print(json.dumps({"groups": G, "replicas": K, "votes": N, "sweep": out}))


if __name__ == "__main__":
    main()
