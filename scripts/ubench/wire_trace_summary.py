#!/usr/bin/env python
"""Summary of a k_wire_decode1 timeline (scripts/ubench/wire_trace.sh): per tile eight wall_clock64 stamps
(100 MHz: 10 ns) of thread 0 - 0 entry, 1 ticket drawn, 2 staged, 7 own lookup begins, 3 own parse + lookup
done, 4 all lanes parsed (counts known), 5 look-back done, 6 records written."""
import sys

import numpy as np

t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16).astype(np.int64)
t = t[(t[:, 0] > 0) & (t[:, 6] > 0)]
t0 = t[:, 0].min()
us = (t - t0) / 100.0
n = us.shape[0]
print(f"tiles {n}  kernel span {us[:, 6].max():.1f} us  (first entry -> last tile written)")
names = [("entry -> ticket", 0, 1), ("ticket -> staged", 1, 2), ("staged -> own lookup begins (own parse)", 2, 7),
         ("own lookup", 7, 3), ("own parse done -> all lanes parsed + scans", 3, 4), ("look-back", 4, 5),
         ("emit", 5, 6), ("whole tile life", 0, 6)]
for name, a, b in names:
    d = us[:, b] - us[:, a]
    print(f"  {name:46s} mean {d.mean():7.2f}  p50 {np.median(d):7.2f}  p90 {np.percentile(d, 90):7.2f}  max {d.max():7.2f} us")
# concurrency: tiles alive over time
ev = np.concatenate([np.stack([us[:, 0], np.ones(n)], 1), np.stack([us[:, 6], -np.ones(n)], 1)])
ev = ev[np.argsort(ev[:, 0])]
alive = np.cumsum(ev[:, 1])
dt = np.diff(ev[:, 0], append=ev[-1, 0])
print(f"  tiles alive: time-weighted mean {float((alive * dt).sum() / max(dt.sum(), 1e-9)):.0f}  max {int(alive.max())}")
order = np.argsort(us[:, 1])
gaps = np.diff(us[order, 1])
print(f"  ticket-to-ticket gap: mean {gaps.mean() * 1000:.1f} ns  p50 {np.median(gaps) * 1000:.1f} ns  (tickets drawn over {us[:, 1].max() - us[:, 1].min():.1f} us)")
q = max(n // 8, 1)
for k in range(0, n, q):
    sel = slice(k, min(k + q, n))
    print(f"  tiles {k:5d}..: entry {us[sel, 0].mean():7.1f}  counts known {us[sel, 4].mean():7.1f}  look-back done {us[sel, 5].mean():7.1f}  written {us[sel, 6].mean():7.1f}")
