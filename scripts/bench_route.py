#!/usr/bin/env python
"""Device-side hash-sharding of one mixed batch (gpx_route_batch_dev): BASELINE config #4's
5 M votes of a 1 M-group, 5-replica round binned for 8 GPUs.  Not the judged bench."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gigapaxos_amd import Engine, load_hip, streams  # noqa: E402
from gigapaxos_amd.sharding import ShardMap  # noqa: E402


def main():
    G, K, NS = 1_000_000, 5, 8
    dev = torch.device("cuda:0")
    cols = streams.vote_round(G, list(range(100, 100 + K)), 0, 100, config_id=4)
    n = cols[0].shape[0]
    e = Engine(load_hip(), 100, 64, kmax=3, window=8, max_batch=n + 16)
    sm = ShardMap(G, NS)
    d_in = [torch.from_numpy(c).to(dev) for c in cols]
    d_out = [torch.empty(n, dtype=torch.int32, device=dev) for _ in cols]
    g2l = torch.from_numpy(sm.local).to(dev)
    off = torch.zeros(NS + 1, dtype=torch.int32, device=dev)
    ts = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(ts)
    e.set_stream(ts.cuda_stream)
    args = (n, [t.data_ptr() for t in d_in], g2l.data_ptr(), G, NS, [t.data_ptr() for t in d_out], off.data_ptr())
    for _ in range(3):
        e.route_dev(*args)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    ev0.record()
    for _ in range(reps):
        e.route_dev(*args)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / reps
    e.profile(2)
    for _ in range(5):
        e.route_dev(*args)
    torch.cuda.synchronize()
    kern = {k: round(v[1] * 1e3 / 5, 1) for k, v in e.profile_read().items()}
    counts = np.diff(off.cpu().numpy())
    assert counts.sum() == n and (counts == np.bincount(sm.shard[cols[0]], minlength=NS)).all()
    print(json.dumps({"votes": int(n), "columns": len(cols), "shards": NS, "ms": round(ms, 4),
                      "votes_per_sec": round(n / ms * 1e3, 1), "GBps_in_plus_out": round(n * 24 * 2 / ms / 1e6, 1),
                      "kernels_us": kern}))


if __name__ == "__main__":
    main()
