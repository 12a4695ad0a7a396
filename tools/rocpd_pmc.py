#!/usr/bin/env python
"""Per-kernel average of every PMC counter in a rocprofv3 rocpd sqlite file."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info('counters_collection')")]
print(cols, file=sys.stderr)
rows = c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
for k, n, v, cnt in rows:
    print("%-40s %-32s %16.1f  (n=%d)" % (k.split("(")[0][-48:], n, v, cnt))
