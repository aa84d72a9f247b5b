#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel in a source file: hipcc -S for gfx950, then the .amdhsa_ directives.
   python tools/kernel_regs.py rsparse_amd/csrc/wrmf_cgq.hip [filter-substring] [-D...]"""
import re
import subprocess
import sys
import tempfile

src = sys.argv[1]
flt = [a for a in sys.argv[2:] if not a.startswith("-D")]
defs = [a for a in sys.argv[2:] if a.startswith("-D")]
with tempfile.TemporaryDirectory() as td:
    out = td + "/k.s"
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", *defs, src, "-o", out],
                          stderr=subprocess.DEVNULL)
    s = open(out).read()
dem = {}
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
    name, body = m.group(1), m.group(2)
    g = lambda key: (re.search(r"\.amdhsa_%s (\d+)" % key, body) or [None, "?"])[1]
    dem[name] = (g("next_free_vgpr"), g("accum_offset"), g("private_segment_fixed_size"), g("group_segment_fixed_size"))
names = list(dem)
pretty = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
for n, p in zip(names, pretty):
    p = re.sub(r"\(.*", "", p.replace("void rsparse_hip::(anonymous namespace)::", ""))
    if flt and not all(f in p for f in flt):
        continue
    v, acc, scr, lds = dem[n]
    print("%-70s vgpr+agpr %4s (arch %4s)  scratch %5s B  static LDS %6s" % (p[:70], v, acc, scr, lds))
