#!/usr/bin/env python3
"""Per-kernel averages of every counter found in a set of rocprofv3 PMC result databases.
usage: pmc_table.py out.txt db1 [db2 ...]   (one database per --pmc pass)"""
import re
import sqlite3
import sys


def short(name):
    m = re.match(r"(?:void )?([A-Za-z_0-9]+(?:<[0-9a-z, ]+>)?)", name)
    return m.group(1) if m else name[:40]


def main():
    out, dbs = sys.argv[1], sys.argv[2:]
    table, durs = {}, {}
    for db in dbs:
        c = sqlite3.connect(db)
        try:
            rows = c.execute(
                "select kernel_name, grid_size, counter_name, count(*), avg(value), avg(duration) "
                "from counters_collection group by kernel_name, grid_size, counter_name").fetchall()
        except sqlite3.Error as e:
            print("skip", db, e)
            continue
        for kn, grid, ctr, cnt, val, dur in rows:
            k = "%s@%d" % (short(kn), grid)
            if not k.startswith("k_"):
                continue
            table.setdefault(k, {})[ctr] = val
            durs[k] = dur / 1e3
    ctrs = sorted({c for v in table.values() for c in v})
    lines = []
    for k in sorted(table, key=lambda k: -durs[k]):
        lines.append("%s   avg %.1f us (under counters)" % (k, durs[k]))
        for c in ctrs:
            if c in table[k]:
                lines.append("    %-40s %16.1f" % (c, table[k][c]))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
