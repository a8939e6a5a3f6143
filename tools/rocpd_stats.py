#!/usr/bin/env python
"""Per-kernel summary of a rocprofv3 (rocpd SQLite) kernel trace:  python tools/rocpd_stats.py results.db [> profiles/x.txt]"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % kd)]
    scols = [r[1] for r in db.execute("pragma table_info(%s)" % ks)]
    name_col = "display_name" if "display_name" in scols else ("kernel_name" if "kernel_name" in scols else "name")
    q = ("select s.%s, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
         "from %s d join %s s on d.kernel_id = s.id group by s.%s order by 3 desc" % (name_col, kd, ks, name_col))
    rows = list(db.execute(q))
    total = sum(r[2] for r in rows)
    print("%-64s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for n, c, t, a, mn, mx in rows:
        n = n.split("(")[0]
        print("%-64s %8d %12.1f %10.2f %10.2f %10.2f %6.2f" % (n[-64:], c, t / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / total))
    print("total kernel time: %.3f ms over %d dispatches" % (total / 1e6, sum(r[1] for r in rows)))
    if "--cols" in sys.argv:
        print(cols)


if __name__ == "__main__":
    main(sys.argv[1])
