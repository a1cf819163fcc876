#!/usr/bin/env python3
"""Turns rocprofv3's rocpd sqlite outputs (gpurun_out/prof_*/..._results.db) into the small text /
json summaries committed under profiles/.

  kernel trace (--kernel-trace --stats)  -> per-kernel calls / total / average duration
  PMC passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; one counter per pass, as
  MI355X_MICROARCH.md §HBM / §rocprofv3 PMC slots prescribes: FETCH_SIZE costs 3 TCC slots and
  WRITE_SIZE 2, they do not fit one pass)            -> per-kernel average KB per launch

HBM-traffic correction (MI355X_MICROARCH.md §HBM): on gfx950 FETCH_SIZE reports exactly 1/2 of
the bytes of a WIDE COALESCED streaming read (16 B/lane); other access widths and WRITE_SIZE are
uncalibrated there.  Round 6 calibrated them on this engine's own access patterns
(scripts/ubench/ubench_counters.hip, profiles/r06_counter_calibration.txt): FETCH_SIZE / true bytes
= 0.5 for every DENSE read (16, 4, 2, 1 bytes per lane), 1.0 for 8-byte entries read as scattered
64-byte lines and for random 4-byte gathers (one line each); WRITE_SIZE / true bytes = 1.0 for every
dense store.  So the read factor is per kernel (READ_FACTOR below): 2 for the streaming kernels;
the tiled per-bucket kernel reads its state columns densely (x2) and its record runs as scattered
lines (x1) - its factor is the one that reproduces the known state bytes (48 B per group) plus the
rest at x1, and both bounds (x1, x2) are kept beside it.
"""
import json
import re
import sqlite3
import sys


# read factor per kernel (prefix match, first hit); default 2.0 = dense reads
READ_FACTOR = [("k_bucket_ar16_tiles", 1.67), ("k_bucket_ar16_slots", 1.5), ("k_ar_tiny", 1.0)]


def read_factor(kernel):
    for prefix, f in READ_FACTOR:
        if kernel.startswith(prefix):
            return f
    return 2.0


def short(name):
    m = re.match(r"(?:void )?([A-Za-z_0-9]+(?:<[0-9a-z, ]+>)?)", name)
    return m.group(1) if m else name[:40]


def kernel_stats(db):
    c = sqlite3.connect(db)
    rows = c.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    return [(short(n), int(calls), float(tot), float(avg), float(pct)) for n, calls, tot, avg, pct in rows]


def counter_avgs(db, counter):
    c = sqlite3.connect(db)
    try:
        rows = c.execute(
            "select kernel_name, grid_size, count(*), avg(value), avg(duration) from counters_collection "
            "where counter_name=? group by kernel_name, grid_size", (counter,)).fetchall()
    except sqlite3.Error:
        return []  # a trace without this PMC pass
    return [(short(n), int(grid), int(cnt), float(val), float(dur)) for n, grid, cnt, val, dur in rows]


def main():
    kt, fetch, write, out_txt, out_json = sys.argv[1:6]
    lines = []
    lines.append("# rocprofv3 --kernel-trace --stats   (durations in us)")
    lines.append("%-28s %8s %12s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for n, calls, tot, avg, pct in kernel_stats(kt):
        if n.startswith(("k_", "__amd")):
            lines.append("%-28s %8d %12.1f %10.2f %6.2f%%" % (n, calls, tot, avg, pct))
    traffic = {}
    for label, db, ctr in (("FETCH_SIZE", fetch, "FETCH_SIZE"), ("WRITE_SIZE", write, "WRITE_SIZE")):
        lines.append("")
        lines.append("# rocprofv3 --kernel-trace --pmc %s   (KB per launch, averaged per kernel and grid size)" % ctr)
        lines.append("%-28s %10s %6s %14s %10s" % ("kernel", "grid", "n", "avg_KB", "avg_us"))
        for n, grid, cnt, val, dur in sorted(counter_avgs(db, ctr)):
            if n.startswith("k_"):
                lines.append("%-28s %10d %6d %14.1f %10.2f" % (n, grid, cnt, val, dur / 1e3))
                traffic.setdefault("%s@%d" % (n, grid), {})[label] = val * 1024.0
    for k, v in traffic.items():
        f, w = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
        v["hbm_bytes_raw"] = f + w
        rf = read_factor(k.split("@")[0])
        v["read_factor"] = rf
        v["hbm_bytes_corrected"] = rf * f + w
        v["hbm_bytes_low"], v["hbm_bytes_high"] = f + w, 2.0 * f + w
    open(out_txt, "w").write("\n".join(lines) + "\n")
    json.dump(traffic, open(out_json, "w"), indent=1, sort_keys=True)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
