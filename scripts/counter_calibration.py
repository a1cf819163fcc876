#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE of scripts/ubench/ubench_counters.hip's kernels over the bytes they really move.
usage: counter_calibration.py true_bytes.json fetch_results.db write_results.db kt_results.db out.txt out.json"""
import json
import sqlite3
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from rocprof_summary import counter_avgs, kernel_stats  # noqa: E402


def main():
    true_json, fetch_db, write_db, kt_db, out_txt, out_json = sys.argv[1:7]
    true = json.load(open(true_json))
    fetch = {n: v * 1024.0 for n, _, _, v, _ in counter_avgs(fetch_db, "FETCH_SIZE")}
    write = {n: v * 1024.0 for n, _, _, v, _ in counter_avgs(write_db, "WRITE_SIZE")}
    avg_us = {n: avg for n, _, _, avg, _ in kernel_stats(kt_db)}  # (top_kernels.average is in microseconds)
    lines = ["# counter calibration on this engine's access patterns (scripts/ubench/ubench_counters.hip), MI355X",
             "# FETCH_SIZE / WRITE_SIZE are rocprofv3's figures (KB x 1024), one counter per pass; true = bytes the kernel moves",
             "%-22s %14s %14s %8s %14s %14s %8s %9s %9s  %s" % ("kernel", "true_read", "FETCH_SIZE", "ratio", "true_write", "WRITE_SIZE", "ratio",
                                                          "avg_us", "TB/s", "pattern")]
    out = {}
    for k, t in true.items():
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        rr = f / t["read"] if t["read"] else None
        wr = w / t["write"] if t["write"] else None
        us = avg_us.get(k)
        tb = (t["read"] + t["write"]) / (us * 1e-6) / 1e12 if us else None
        out[k] = {"true_read": t["read"], "FETCH_SIZE": f, "fetch_ratio": rr, "true_write": t["write"], "WRITE_SIZE": w,
                  "write_ratio": wr, "avg_us": us, "tb_per_s": tb, "what": t["what"]}
        for extra in ("sectors32_bytes", "lines64_bytes"):
            if extra in t:
                out[k]["fetch_over_" + extra] = f / t[extra]
        lines.append("%-22s %14d %14.0f %8s %14d %14.0f %8s %9s %9s  %s" % (
            k, t["read"], f, "-" if rr is None else "%.3f" % rr, t["write"], w, "-" if wr is None else "%.3f" % wr,
            "-" if us is None else "%.1f" % us, "-" if tb is None else "%.2f" % tb, t["what"]))
        if "sectors32_bytes" in t:
            lines.append("%-22s   FETCH_SIZE / (32 B per gather) = %.3f, / (64 B per gather) = %.3f; the other kernel's FETCH_SIZE under a "
                         "write-only kernel is the counter's floor" % ("", f / t["sectors32_bytes"], f / t["lines64_bytes"]))
    lines.append("# (a write-only kernel's FETCH_SIZE and a read-only kernel's WRITE_SIZE:)")
    for k, t in true.items():
        lines.append("#   %-22s FETCH_SIZE %12.0f  WRITE_SIZE %12.0f" % (k, fetch.get(k, 0.0), write.get(k, 0.0)))
    open(out_txt, "w").write("\n".join(lines) + "\n")
    json.dump(out, open(out_json, "w"), indent=1, sort_keys=True)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
