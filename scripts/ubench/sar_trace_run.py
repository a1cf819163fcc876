#!/usr/bin/env python
"""Runs a few tiny / small accept-reply calls on the -DGPX_SAR_TRACE build and prints, per kernel stamp, when the
workgroups of the LAST call of each shape passed it (us after the first workgroup's entry)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gigapaxos_amd import Engine, hri_create, load_hip, streams, S_OK, ORDERED_REPLY_RUNS, LAZY_OUTPUTS  # noqa: E402

TRACE = "/tmp/sar_trace.bin"
os.environ["GPX_SAR_TRACE_FILE"] = TRACE
NAMES = {"k_ar_tiny": ["entry", "", "", "votes loaded", "placed", "thread 0 replayed", "all replayed", "", "outputs written"],
         "k_ar_runs_small": ["entry", "verdict exchanged", "lane 0 straight-line done", "all lanes done"]}


def summary(kernel):
    tr = np.fromfile(TRACE, dtype=np.uint64).reshape(256, 16).astype(np.int64)
    live = tr[:, 0] > 0
    tr = tr[live]
    t0 = tr[:, 0].min()
    print(f"  {kernel}: {tr.shape[0]} workgroups; entries spread over {(tr[:, 0].max() - t0) / 100.0:.2f} us")
    for k, nm in enumerate(NAMES[kernel]):
        col = tr[:, k]
        col = col[col > 0]
        if col.size:
            us = (col - t0) / 100.0  # wall_clock64: 100 MHz
            print(f"    {k} {nm:28s} min {us.min():7.2f}  median {np.median(us):7.2f}  max {us.max():7.2f} us")


def main():
    dev = torch.device("cuda:0")
    P = lambda t: t.data_ptr()  # noqa: E731
    for G, n, runs in ((1_000_000, 1024, False), (10_000, 1024, False), (10_000, 30_000, True), (1_000_000, 256, False)):
        K, members = 3, [100, 101, 102]
        e = Engine(load_hip(), 100, G, kmax=K, window=8, max_batch=max(n, G) + 1024)
        mem = np.tile(np.array(members, np.int32), (G, 1))
        assert (e.create_groups(np.arange(G, dtype=np.int32), mem, K, hri_create(G, K, 100)) == S_OK).all()
        if runs:
            e.set_ordered_batches(ORDERED_REPLY_RUNS | LAZY_OUTPUTS)
        live = np.sort(np.random.default_rng(1).choice(G, min(G, n // K), replace=False)).astype(np.int32)
        d = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(5)] + [torch.zeros(n, dtype=torch.uint8, device=dev)]
        no, st = torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.uint8, device=dev)
        for r in range(4):
            e.propose(live)
            cols = (streams.vote_round_runs if runs else streams.vote_round)(G, members, r, 100, groups=live)
            nn = min(n, cols[0].shape[0])
            dc = [torch.from_numpy(np.ascontiguousarray(c[:nn])).to(dev) for c in cols]
            torch.cuda.synchronize()
            e.call_dev("accept_reply_batch", nn, *[P(c) for c in dc], *[P(t) for t in d], P(no), P(st))
            torch.cuda.synchronize()
        print(f"G = {G}, {nn} votes, {'ascending runs' if runs else 'shuffled'}: n_out = {int(no)}")
        summary("k_ar_runs_small" if runs else "k_ar_tiny")
        e.close()


if __name__ == "__main__":
    main()
