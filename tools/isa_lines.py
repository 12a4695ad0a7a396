#!/usr/bin/env python3
"""Static instruction count of a kernel per source line (hipcc -gline-tables-only; no GPU needed): where a kernel's code
size comes from.  python tools/isa_lines.py <file.hip> <mangled-name substring> [top N]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, pat = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                    "-gline-tables-only", "-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"), src, "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().splitlines()
files = {}
for ln in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
start = next(i for i, ln in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pat), ln))
cnt, cur = collections.Counter(), None
for ln in lines[start:]:
    if ".Lfunc_end" in ln:
        break
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", ln)
    if m:
        cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
        continue
    s = ln.split(";")[0].strip()
    if s and not s.endswith(":") and not s.startswith("."):
        cnt[cur] += 1
print("total instructions", sum(cnt.values()))
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1])[:top]:
    print("%6d  %s:%s" % (v, k[0] if k else "?", k[1] if k else "?"))
