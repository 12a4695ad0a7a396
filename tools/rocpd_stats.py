#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / min / max (us).
Usage: python tools/rocpd_stats.py <results.db> [out.csv]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["kernel,calls,total_us,avg_us,min_us,max_us,percent"]
    for name, n, s, a, mn, mx in rows:
        short = name.split("(")[0]
        lines.append('"%s",%d,%.1f,%.2f,%.2f,%.2f,%.2f' % (short, n, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
