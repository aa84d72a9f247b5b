#!/usr/bin/env python3
"""Instruction mix of every loop of one kernel in a device assembly listing (no GPU): per loop, how many MFMA / packed and
plain fp VALU / DPP / cross-lane / v_mov / LDS / global / scalar / s_nop / s_waitcnt instructions its body holds.  Says what a
sweep is made of before a counter run does (e.g. the 257-512 bucket's sweep: 76 v_pk_fma + 40 DPP adds + 57 hazard s_nops
in 390 instructions).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only rsparse_amd/csrc/wrmf_cgq.hip -o /tmp/cgq.s
    python tools/dbg/loop_mix.py /tmp/cgq.s als_cgq_kernelILi128ELi16ELi8ELi8ELi0ELb1ELi0ELb0E

Companion of tools/dbg/spill_sites.py and tools/kernel_resources.py."""
import collections
import re
import sys


def kind(line):
    op = line.split()[0]
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("v_permlane", "v_readlane", "v_readfirstlane", "v_writelane", "v_bpermute", "v_swap")):
        return "xlane"
    if "dpp" in line:
        return "dpp"
    if op.startswith(("v_mov", "v_accvgpr")):
        return "v_mov"
    if op.startswith("v_pk_"):
        return "pk_fp"
    if op.startswith(("v_fma", "v_fmac", "v_mul_f", "v_add_f", "v_sub_f", "v_mac", "v_dot")):
        return "fp"
    return "valu_other" if op.startswith("v_") else "other"


def main():
    listing, pat = sys.argv[1], sys.argv[2]
    s = open(listing).read().splitlines()
    i0 = [i for i, l in enumerate(s) if re.match(r"_ZN\S*" + pat + r"\S*:", l)][0]
    i1 = next(i for i in range(i0, len(s)) if s[i].startswith(".Lfunc_end"))
    body = s[i0:i1]
    ins = [(i, l.strip()) for i, l in enumerate(body) if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    label_at = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"\.LBB\d+_\d+:", l)}
    loops = []
    for i, l in ins:
        m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and label_at.get(m.group(1), i) < i:
            loops.append((label_at[m.group(1)], i, m.group(1)))
    print(s[i0].split(":")[0][:110], "--", len(ins), "instructions,", len(loops), "loops")
    for a, b, t in sorted(loops):
        sub = [l for i, l in ins if a <= i <= b]
        nested = sum(1 for x in loops if a < x[0] and x[1] < b)
        c = collections.Counter(kind(l) for l in sub)
        print("  %-12s %5d instructions, %2d loops inside:  %s" % (t, len(sub), nested, "  ".join("%s %d" % kv for kv in sorted(c.items(), key=lambda kv: -kv[1]))))


if __name__ == "__main__":
    main()
