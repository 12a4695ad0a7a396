#!/usr/bin/env python
"""Per-kernel average duration for each phase of tools/step_timeline.py (phases are separated by a cos kernel)."""
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
phase = 0
acc = defaultdict(lambda: [0, 0.0])
for name, s, e in rows:
    if "cos_kernel" in name:
        phase += 1
        continue
    if "ddsp::" not in name:
        continue
    k = (phase, name.split("(")[0].replace("void ", ""))
    acc[k][0] += 1
    acc[k][1] += (e - s) / 1e3
for (ph, name), (cnt, tot) in sorted(acc.items()):
    print("phase %d  %-36s n=%4d  avg %7.2f us" % (ph, name, cnt, tot / cnt))
