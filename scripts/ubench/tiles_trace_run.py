#!/usr/bin/env python
"""Runs a few shuffled accept-reply rounds on the -DGPX_TL_TRACE build and prints, per kernel stamp, when the workgroups
of the LAST call passed it (us after the first workgroup's entry) and how long each phase took per workgroup."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gigapaxos_amd import Engine, hri_create, load_hip, streams, S_OK  # noqa: E402

TRACE = "/tmp/tl_trace.bin"
os.environ["GPX_TL_TRACE_FILE"] = TRACE
SCATTER = ["entry", "columns loaded, counted", "scanned, A.off written", "sorted in LDS", "stores issued"]
BUCKET = ["entry", "rows read + scanned", "records fetched, counted", "placed", "thread 0 replayed", "outputs staged"]


def summary(title, tr, names):
    tr = tr[tr[:, 0] > 0]
    if not tr.shape[0]:
        print(f"  {title}: no workgroup stamped")
        return
    t0 = tr[:, 0].min()
    last = len(names) - 1
    print(f"  {title}: {tr.shape[0]} workgroups; entries spread over {(tr[:, 0].max() - t0) / 100.0:.2f} us; "
          f"last exit {(tr[:, last].max() - t0) / 100.0:.2f} us; a workgroup lives {np.median(tr[:, last] - tr[:, 0]) / 100.0:.2f} us (median)")
    for k, nm in enumerate(names):
        col = tr[:, k]
        ok = col > 0
        if not ok.any():
            continue  # (a stamp this path does not pass)
        us = (col[ok] - t0) / 100.0  # wall_clock64: 100 MHz
        line = f"    {k} {nm:28s} at min {us.min():7.2f}  median {np.median(us):7.2f}  max {us.max():7.2f} us"
        prev = k - 1
        while prev > 0 and not (tr[:, prev] > 0).any():
            prev -= 1
        if k:
            d = (tr[ok, k] - tr[ok, prev]) / 100.0
            line += f"   phase: median {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f}"
        print(line)


def main():
    dev = torch.device("cuda:0")
    P = lambda t: t.data_ptr()  # noqa: E731
    shapes = [(1_000_000, 3), (1_000_000, 5), (125_000, 5)]
    if len(sys.argv) > 2:
        shapes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
    for G, K in shapes:
        members = list(range(100, 100 + K))
        n = G * K
        e = Engine(load_hip(), 100, G, kmax=K, window=8, max_batch=n + 4096)
        mem = np.tile(np.array(members, np.int32), (G, 1))
        assert (e.create_groups(np.arange(G, dtype=np.int32), mem, K, hri_create(G, K, 100)) == S_OK).all()
        g = np.arange(G, dtype=np.int32)
        d = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(5)] + [torch.zeros(n, dtype=torch.uint8, device=dev)]
        no, st = torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.uint8, device=dev)
        for r in range(3):
            e.propose(g)
            cols = streams.vote_round(G, members, r, 100, config_id=3 if K == 3 else 4)
            dc = [torch.from_numpy(c).to(dev) for c in cols]
            torch.cuda.synchronize()
            e.call_dev("accept_reply_batch", n, *[P(c) for c in dc], *[P(t) for t in d], P(no), P(st))
            torch.cuda.synchronize()
        print(f"G = {G}, K = {K}, {n} shuffled votes: n_out = {int(no)}  (GPX_TILE_T={os.environ.get('GPX_TILE_T', 'auto')})")
        tr = np.fromfile(TRACE, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
        sc = tr[:4096]
        sc = sc[sc[:, 0] > 0]
        if sc.shape[0] and (sc[:, 7] > sc[:, 6]).all():
            mhz = (sc[:, 7] - sc[:, 6]) / ((sc[:, 4] - sc[:, 0]) / 100.0)
            print(f"  shader clock over the scatter workgroups' lives: median {np.median(mhz):.0f} MHz (min {mhz.min():.0f}, max {mhz.max():.0f})")
        summary("k_scatter_tiles", tr[:4096], SCATTER)
        summary("k_bucket_ar16_tiles", tr[4096:], BUCKET)
        e.close()


if __name__ == "__main__":
    main()
