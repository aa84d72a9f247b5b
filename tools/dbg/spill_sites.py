#!/usr/bin/env python3
"""Where a kernel's scratch traffic sits: for every scratch_load / scratch_store of a kernel in a device assembly listing,
the two innermost loops (backward branches) that contain it, with their lengths in lines.  A spill in the per-row loop of
10 000 lines is paid once per row; one inside a 300-line sweep is paid every sweep.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only rsparse_amd/csrc/wrmf_chol_wave.hip -o /tmp/cw.s
    python tools/dbg/spill_sites.py /tmp/cw.s als_chol_wave_kernelILi64ELb1E        # (a piece of the mangled name)

Companion of tools/kernel_resources.py, which says WHICH kernels spill."""
import re
import sys


def main():
    listing, pat = sys.argv[1], sys.argv[2]
    s = open(listing).read().splitlines()
    for i0 in [i for i, l in enumerate(s) if re.match(r"_ZN\S*" + pat + r"\S*:", l)]:
        i1 = next(i for i in range(i0, len(s)) if s[i].startswith(".Lfunc_end"))
        body = s[i0:i1]
        print(s[i0].split(":")[0][:110], "--", len(body), "lines")
        label_at = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"\.LBB\d+_\d+:", l)}
        loops = []
        for i, l in enumerate(body):
            m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
            if m and label_at.get(m.group(1), i) < i:
                loops.append((label_at[m.group(1)], i, m.group(1)))
        loops.sort(key=lambda x: x[1] - x[0])
        sites = {}
        for i, l in enumerate(body):
            if "scratch_" in l:
                inner = tuple("%s[%d]" % (t, b - a) for a, b, t in loops if a <= i <= b)[:2]
                key = (inner, l.split()[0])
                sites[key] = sites.get(key, 0) + 1
        print("  %d loops; scratch instructions by (innermost loops [lines]), opcode:" % len(loops))
        for k, v in sorted(sites.items()):
            print("   %4d  %-26s %s" % (v, k[1], " in ".join(k[0]) or "(outside every loop)"))


if __name__ == "__main__":
    main()
