#!/usr/bin/env python3
"""Loads that are followed AT ONCE by `s_waitcnt vmcnt(0)` in a kernel of a device listing (no GPU): dependent round trips the
compiler serialised -- e.g. a per-lane `cond ? load : 0` whose dependent address arithmetic it sank into the branch (round 5: eight
index round trips per pair of rows in wrmf_cgp.hip, 64 Gramian loads at every DMF workgroup start).  Per hit: the line and the loop.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -Iinclude rsparse_amd/csrc/wrmf_cgp.hip -o /tmp/cgp.s
    python tools/dbg/serial_loads.py /tmp/cgp.s als_cgp_kernelILb0ELb1E"""
import re, sys
f, pat = sys.argv[1], sys.argv[2]
s = open(f).read().splitlines()
i0 = [i for i, l in enumerate(s) if l.startswith("_ZN") and pat in l and (":" in l and not l.startswith("\t"))][0]
i1 = next(i for i in range(i0, len(s)) if s[i].startswith(".Lfunc_end"))
body = s[i0:i1]
ins = [(i, l.strip()) for i, l in enumerate(body) if l.startswith("\t") and not l.strip().startswith((".", ";"))]
hits = []
for n, (i, l) in enumerate(ins):
    if l.startswith(("global_load", "buffer_load")):
        for (j, m) in ins[n + 1:n + 4]:
            if m.startswith("s_waitcnt") and "vmcnt(0)" in m:
                hits.append((i, l.split()[0]))
                break
            if m.startswith(("global_load", "buffer_load")):
                break
# loop headers
loops = [(i, l) for i, l in enumerate(body) if re.match(r"\.LBB\d+_\d+:.*Loop", l)]
print(pat, "loads followed at once by vmcnt(0):", len(hits))
for i, op in hits:
    hdr = [l for j, l in loops if j <= i]
    print("  line", i, op, "|", (hdr[-1].split(";")[-1].strip() if hdr else "prologue"))
