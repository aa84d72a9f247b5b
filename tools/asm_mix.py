#!/usr/bin/env python3
"""Instruction mix of the als_cgq_kernel instantiations in a gfx950 .s file (hipcc -save-temps)."""
import collections
import re
import sys

s = open(sys.argv[1]).read()
want_kp = sys.argv[2] if len(sys.argv) > 2 else "128"
parts = re.split(r"\n\t\.globl\t", s)
for f in parts[1:]:
    name = f.split()[0]
    m = re.search(r"als_cgq_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)", name)
    if not m or m.group(1) != want_kp:
        continue
    body = f.split(".Lfunc_end")[0]
    c = collections.Counter(re.findall(r"^\s+([a-z_0-9]+)", body, re.M))
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    vg = re.search(r"\.vgpr_count:\s+(\d+)", f)
    print("KP,CAPQ,W,WPR,STREAM,IMPL=%s" % ",".join(m.groups()), "VALU", valu, "pk_fma", c["v_pk_fma_f32"], "fma",
          c["v_fma_f32"] + c["v_fmac_f32"], "pk_mul", c["v_pk_mul_f32"], "pk_add", c["v_pk_add_f32"], "mov",
          c["v_mov_b32"] + c["v_accvgpr_read_b32"] + c["v_accvgpr_write_b32"], "dpp", len(re.findall(r"_dpp", body)),
          "ds_read", sum(v for k, v in c.items() if k.startswith("ds_read")), "scratch",
          sum(v for k, v in c.items() if k.startswith("scratch")), "cndmask", c["v_cndmask_b32"],
          "global_load", sum(v for k, v in c.items() if k.startswith("global_load")))
