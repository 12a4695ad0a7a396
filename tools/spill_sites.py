#!/usr/bin/env python3
"""Where a kernel spills: scratch stores / loads per source line (hipcc -gline-tables-only, no GPU needed).
    python tools/spill_sites.py <file.hip> <mangled-name substring> [extra hipcc flags]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, pat, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                    "-gline-tables-only", "-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "ddsp_svc_amd", "csrc"), *extra, src, "-o", out], check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().splitlines()
files = {}
for ln in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
start = next(i for i, ln in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pat), ln))
st, ld, cur, total = collections.Counter(), collections.Counter(), None, 0
for ln in lines[start:]:
    if ".Lfunc_end" in ln:
        break
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", ln)
    if m:
        cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
        continue
    s = ln.split(";")[0].strip()
    if s and not s.endswith(":") and not s.startswith("."):
        total += 1
    if s.startswith("scratch_store"):
        st[cur] += 1
    if s.startswith("scratch_load"):
        ld[cur] += 1
for ln in lines[start:]:
    m = re.search(r"\.(vgpr_count|vgpr_spill_count|sgpr_count):\s+(\d+)", ln)
    if m:
        print(m.group(1), m.group(2))
    if ".Lfunc_end" in ln:
        break
print("instructions", total, "scratch stores", sum(st.values()), "loads", sum(ld.values()))
for k, v in sorted(st.items(), key=lambda kv: -kv[1]):
    print("  store %3d  %s:%s" % (v, k[0] if k else "?", k[1] if k else "?"))
for k, v in sorted(ld.items(), key=lambda kv: -kv[1]):
    print("  load  %3d  %s:%s" % (v, k[0] if k else "?", k[1] if k else "?"))
