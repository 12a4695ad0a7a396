#!/usr/bin/env python
"""Timeline of one steady-state step from a rocprofv3 rocpd sqlite kernel trace: per kernel start offset, duration
and the idle gap before it.  Usage: python tools/rocpd_gaps.py <results.db> [first_kernel_substring]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
first = sys.argv[2] if len(sys.argv) > 2 else "k_phase_frame_sums"
rows = c.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if first in r[0]]
if len(idx) < 3:
    sys.exit("not enough steps in trace")
a, b = idx[-2], idx[-1]                      # the last complete step
t0 = rows[a][1]
prev_end = None
busy = 0
for name, s, e in rows[a:b]:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%9.1f us  dur %8.1f us  gap %6.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, name.split("(")[0][-50:]))
    prev_end = e
    busy += e - s
span = rows[b][1] - t0
print("step span %.1f us, kernels busy %.1f us, idle %.1f us" % (span / 1e3, busy / 1e3, (span - busy) / 1e3))
