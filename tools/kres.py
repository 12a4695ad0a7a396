#!/usr/bin/env python3
"""Registers, spills, scratch, LDS and occupancy of every kernel of a translation unit (hipcc remarks; no GPU needed).
    python tools/kres.py ddsp_svc_amd/csrc/loss_czt.hip [extra hipcc flags]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, extra = sys.argv[1], sys.argv[2:]
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                    "--cuda-device-only", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "ddsp_svc_amd", "csrc"),
                    "-Rpass-analysis=kernel-resource-usage", *extra, "-c", src, "-o", "/dev/null"], capture_output=True, text=True)
cur = None
rows = {}
for ln in r.stderr.splitlines():
    m = re.search(r"remark:\s+Function Name: (\S+)", ln)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", ln)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
if r.returncode:
    print(r.stderr[-2000:])
for name, d in rows.items():
    try:
        nice = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    except Exception:
        nice = name
    print("%-60s vgpr %3d spill %3d scratch %4d lds %6d occ %d sgpr %3d" % (nice[-60:], d.get("VGPRs", -1), d.get("VGPRs Spill", -1),
          d.get("ScratchSize", -1), d.get("LDS Size", -1), d.get("Occupancy", -1), d.get("TotalSGPRs", -1)))
