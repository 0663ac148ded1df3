#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (or a kernel_stats CSV) into a per-kernel table:
calls, total ms, average us, share.  Usage: tools/rocprof_summary.py <results.db> [top_n]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return name if len(name) <= 110 else name[:107] + "..."


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = cur.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
                       f"from {disp} d join {sym} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"{'kernel':110s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'share':>6s}")
    for name, n, t, mn, mx in rows[:top]:
        print(f"{short(name):110s} {n:7d} {t / 1e6:10.3f} {t / n / 1e3:10.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100 * t / total:5.1f}%")
    print(f"{'TOTAL':110s} {sum(r[1] for r in rows):7d} {total / 1e6:10.3f}")


if __name__ == "__main__":
    main()
