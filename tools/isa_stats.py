#!/usr/bin/env python3
"""Instruction mix of a kernel's main loop from hipcc's gfx950 assembly (no GPU needed).

    python tools/isa_stats.py ddsp_svc_amd/csrc/fir_blk.hip 'k_fir_blkILi2ELb0' [--sections]

Compiles the translation unit to assembly (device pass only), cuts out the kernel whose mangled name contains the
pattern, takes the instructions between the loop header label that hipcc marks ("=>This Inner Loop Header" /
"Loop Header") with the largest body and the last branch back to it, and counts them by class.  Conditional
side blocks inside the loop (flushes, stamps) are counted as written, so the figure is an upper bound of one pass.
With --sections the loop is also cut at its s_barrier instructions.  Diagnostics only.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math"]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def classify(op):
    if op.startswith("v_pk_"):
        return "valu_packed"
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op in ("v_sin_f32", "v_cos_f32", "v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32",
              "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64"):
        return "valu_trans"
    if op.endswith("_dpp") or "permlane" in op or op.startswith("v_readlane") or op.startswith("v_writelane") \
            or op.startswith("v_readfirstlane"):
        return "valu_xlane"
    if op.endswith("_f64") or "_f64_" in op:
        return "valu_f64"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "lds_read"
    if op.startswith("ds_write") or op.startswith("ds_store"):
        return "lds_write"
    if op.startswith("ds_"):
        return "lds_other"
    if op.startswith("buffer_load") or op.startswith("global_load") or op.startswith("flat_load") or op.startswith("scratch_load"):
        return "vmem_load"
    if op.startswith("buffer_store") or op.startswith("global_store") or op.startswith("flat_store") or op.startswith("scratch_store"):
        return "vmem_store"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_"):
        return "vmem_other"
    if op == "s_barrier":
        return "barrier"
    if op == "s_waitcnt":
        return "waitcnt"
    if op == "s_nop":
        return "nop"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_cbranch") or op == "s_branch":
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def kernel_text(asm, pattern):
    lines = asm.splitlines()
    start = None
    for i, ln in enumerate(lines):
        m = re.match(r"^(_Z\w+):", ln)
        if m and pattern in m.group(1):
            start = i
            break
    if start is None:
        raise SystemExit("no kernel matching %r" % pattern)
    end = len(lines)
    for i in range(start + 1, len(lines)):
        if lines[i].startswith("\t.section") or re.match(r"^\s*s_endpgm", lines[i]):
            end = i + 1
            if "s_endpgm" in lines[i]:
                # keep going: side blocks may follow the first s_endpgm
                continue
            break
        if re.match(r"^\.Lfunc_end", lines[i]):
            end = i
            break
    return lines[start:end]


def instr(ln):
    s = ln.split(";")[0].strip()
    if not s or s.endswith(":") or s.startswith("."):
        return None
    return s.split()[0]


def main_loop(lines):
    """(header label, [line indices]) of the loop with the most instructions: hipcc tags every block of a loop with
    its header ("in Loop: Header=BBx_y"), the header itself with "Loop Header"."""
    blocks = []                                   # (label, comment, first line, last line + 1)
    cur = None
    for i, ln in enumerate(lines):
        m = re.match(r"^\.L(BB\d+_\d+):(.*)", ln)
        if m:
            if cur:
                blocks.append((cur[0], cur[1], cur[2], i))
            cur = (m.group(1), m.group(2), i)
    if cur:
        blocks.append((cur[0], cur[1], cur[2], len(lines)))
    loops = {}
    for label, comment, a, b in blocks:
        if "Loop Header" in comment:
            loops.setdefault(label, []).extend(range(a, b))
        m = re.search(r"Header=(BB\d+_\d+)", comment)
        if m:
            loops.setdefault(m.group(1), []).extend(range(a, b))
    if not loops:
        return None
    return max(loops.items(), key=lambda kv: len(kv[1]))


def count(lines):
    c = collections.Counter()
    for ln in lines:
        op = instr(ln)
        if op:
            c[classify(op)] += 1
    return c


def _regs(tok):
    """set of VGPR numbers named by an operand token like v12 or v[12:13]"""
    m = re.match(r"^-?\|?v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"^-?\|?v(\d+)\b", tok)
    return {int(m.group(1))} if m else set()


def dependent_adjacent(lines):
    """vector instructions that read a register the instruction right before them wrote (they wait ~3 cycles for it;
    valu_probe2.hip) -- counted over consecutive instructions of the listing, s_nop and s_waitcnt in between ignored"""
    n = 0
    prev_dst = set()
    for ln in lines:
        op = instr(ln)
        if not op or op in ("s_nop", "s_waitcnt"):
            continue
        body = ln.split(";")[0].strip()
        toks = [t.strip() for t in body[len(op):].split(",")]
        if op.startswith("v_") and toks:
            srcs = set()
            for t in toks[1:]:
                srcs |= _regs(t.split()[0] if t else "")
            if op.startswith("v_fmac") or op.startswith("v_mac"):
                srcs |= _regs(toks[0])
            if srcs & prev_dst:
                n += 1
            prev_dst = _regs(toks[0].split()[0])
        else:
            prev_dst = set()
    return n


def fmt(c):
    valu = sum(v for k, v in c.items() if k.startswith("valu"))
    lds = sum(v for k, v in c.items() if k.startswith("lds"))
    keys = sorted(c)
    return "VALU %4d  LDS %3d  | " % (valu, lds) + "  ".join("%s=%d" % (k, c[k]) for k in keys)


def main():
    src, pat = sys.argv[1], sys.argv[2]
    extra = [a for a in sys.argv[3:] if a.startswith("-D")]
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"),
                        src, "-o", out], check=True, stderr=subprocess.DEVNULL)
        asm = open(out).read()
    lines = kernel_text(asm, pat)
    for key in (".vgpr_count", ".vgpr_spill_count", ".sgpr_count", ".group_segment_fixed_size"):
        for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n){0,20}?", asm):
            pass
    # resource lines of the matching kernel from the metadata block
    md = re.search(r"\.name:\s+(_Z\w*" + re.escape(pat) + r"\w*)\n((?:\s+\.\w+:.*\n)+)", asm)
    meta = asm[asm.find("amdhsa.kernels"):]
    blocks = meta.split("  - .agpr_count")
    for b in blocks:
        if pat in b and ".vgpr_count" in b:
            g = lambda k: re.search(r"\." + k + r":\s+(\d+)", b)
            print("vgpr %s  spill %s  sgpr %s  lds %s B" % tuple(g(k).group(1) if g(k) else "?" for k in
                  ("vgpr_count", "vgpr_spill_count", "sgpr_count", "group_segment_fixed_size")))
            break
    print("whole kernel:", fmt(count(lines)))
    ml = main_loop(lines)
    if not ml:
        print("no loop found")
        return
    body = [lines[i] for i in sorted(ml[1])]
    print("main loop   :", fmt(count(body)), " dep-adjacent VALU", dependent_adjacent(body))
    if "--sections" in sys.argv:
        cuts = [i for i, ln in enumerate(body) if instr(ln) == "s_barrier"]
        prev = 0
        for n, cpos in enumerate(cuts + [len(body)]):
            print("  section %d  :" % n, fmt(count(body[prev:cpos])), " dep-adj", dependent_adjacent(body[prev:cpos]))
            prev = cpos


if __name__ == "__main__":
    main()
