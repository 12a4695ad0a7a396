#!/usr/bin/env python
"""Per launch POSITION within a step (the i-th launch after each k_phase_frame_sums), averaged over the traced steps: which of
the three tap syntheses / filters of a CombSub step costs what.  Usage: python tools/rocpd_launches.py <results.db> [first]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
first = sys.argv[2] if len(sys.argv) > 2 else "k_phase_frame_sums"
rows = c.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if first in r[0]]
if len(idx) < 4:
    sys.exit("not enough steps in trace")
steps = [rows[a:b] for a, b in zip(idx[1:-1], idx[2:])]      # skip the first (cold) step
n = min(len(s) for s in steps)
steps = [s for s in steps if len(s) == n]
import os
if os.environ.get("LAST"):                                    # only the last N steps: the clocks' steady state (the first ~100 steps of a
    steps = steps[-int(os.environ["LAST"]):]                  # process are a transient: a filter launch goes 77 -> 100 -> 72 us, r06_v6_*)
print("%d steps of %d launches" % (len(steps), n))
for i in range(n):
    d = sorted((s[i][2] - s[i][1]) / 1e3 for s in steps)
    off = sorted((s[i][1] - s[0][1]) / 1e3 for s in steps)
    print("  #%d  start %7.1f us  dur avg %7.2f  med %7.2f  min %7.2f  %s" % (i, off[len(off) // 2], sum(d) / len(d), d[len(d) // 2], d[0],
          steps[0][i][0].split("(")[0][-44:]))
span = sorted((b[0][1] - a[0][1]) / 1e3 for a, b in zip(steps[:-1], steps[1:]))
if span:
    print("step to step: med %.1f us  min %.1f us" % (span[len(span) // 2], span[0]))
if len(sys.argv) > 3:                                         # time series of one launch position over the traced steps
    i = int(sys.argv[3])
    print("position #%d over the steps:" % i, " ".join("%.1f" % ((s[i][2] - s[i][1]) / 1e3) for s in steps))
    print("step to step over the steps:", " ".join("%.0f" % ((b[0][1] - a[0][1]) / 1e3) for a, b in zip(steps[:-1], steps[1:])))
if os.environ.get("CSV_OUT"):                                 # per-kernel table over the SAME steps (steady state when LAST is set)
    per = {}
    for st_ in steps:
        for name, a_, b_ in st_:
            per.setdefault(name.split("(")[0], []).append((b_ - a_) / 1e3)
    tot = sum(sum(v) for v in per.values()) or 1.0
    with open(os.environ["CSV_OUT"], "w") as f:
        f.write("# the last %d steps of the trace (past the clocks' transient), %d launches per step\n" % (len(steps), n))
        f.write("kernel,calls,total_us,avg_us,min_us,max_us,percent\n")
        for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            f.write('"%s",%d,%.1f,%.2f,%.2f,%.2f,%.2f\n' % (k, len(v), sum(v), sum(v) / len(v), min(v), max(v), 100.0 * sum(v) / tot))
