#!/usr/bin/env python
"""Merge the FETCH_SIZE / WRITE_SIZE passes of tools/gpu_traffic.sh into per-kernel HBM bytes per launch and, with
``--steps N`` (the number of steps the profiled command ran), the bytes of one whole step under ``__step__``.
FETCH_SIZE and WRITE_SIZE are KiB; FETCH_SIZE is doubled (gfx950 tallies 128-B read requests at 64 B,
MI355X_MICROARCH.md section HBM) -- exact for the wide coalesced streams these kernels read."""
import argparse
import json
import re

ap = argparse.ArgumentParser()
ap.add_argument("--model", default=None)
ap.add_argument("--steps", type=int, default=0)
ap.add_argument("paths", nargs="+")
a = ap.parse_args()

out = {}
for path in a.paths:
    for line in open(path):
        m = re.match(r"\s*(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+([0-9.]+)\s+\(n=(\d+)\)", line)
        if not m:
            continue
        k = m.group(1).split("ddsp::")[-1].strip()
        d = out.setdefault(k, {})
        d[m.group(2)] = float(m.group(3))
        d["launches"] = int(m.group(4))
res = {}
for k, d in out.items():
    rd = d.get("FETCH_SIZE", 0.0) * 1024 * 2
    wr = d.get("WRITE_SIZE", 0.0) * 1024
    res[k] = {"read_bytes": rd, "write_bytes": wr, "hbm_bytes": rd + wr, "launches_sampled": d.get("launches"),
              "raw_FETCH_SIZE_KiB": d.get("FETCH_SIZE"), "raw_WRITE_SIZE_KiB": d.get("WRITE_SIZE")}
if a.steps > 0:
    per_step = {k: v["launches_sampled"] / a.steps for k, v in res.items() if k != "k_ir_table"}     # the table is built once
    res["__step__"] = {"model": a.model, "steps_profiled": a.steps,
                       "hbm_bytes": sum(res[k]["hbm_bytes"] * n for k, n in per_step.items()),
                       "read_bytes": sum(res[k]["read_bytes"] * n for k, n in per_step.items()),
                       "write_bytes": sum(res[k]["write_bytes"] * n for k, n in per_step.items()),
                       "launches_per_step": per_step}
print(json.dumps(res, indent=1))
