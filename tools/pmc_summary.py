#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSVs (counter_collection.csv) per kernel name: mean per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = r.get("Kernel_Name", "")
            if "rsparse" not in name:
                continue
            import re
            m = re.search(r"(als_\w+|gramian_\w+|sum_partials_kernel|trace_kernel|weighted_\w+)(<[^>]*>)?", name)
            short = (m.group(1) + (m.group(2) or "")) if m else name[:60]
            agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    d = {c: sum(v) / len(v) for c, v in agg[k].items()}
    print(k, "dispatches", len(next(iter(agg[k].values()))))
    for c in sorted(d):
        print("   %-28s %.4g" % (c, d[c]))
    if "SQ_WAVE_CYCLES" in d and d["SQ_WAVE_CYCLES"] > 0 and "SQ_WAVES" in d:
        wc, w = d["SQ_WAVE_CYCLES"], d["SQ_WAVES"]
        print("   -> of wave-cycles: wait_any %.2f wait_inst %.2f active %.2f | per wave: valu %.0f lds %.0f vmem_rd %.0f vmem_wr %.0f salu %.0f"
              % (d.get("SQ_WAIT_ANY", 0) / wc, d.get("SQ_WAIT_INST_ANY", 0) / wc, d.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                 d.get("SQ_INSTS_VALU", 0) / w, d.get("SQ_INSTS_LDS", 0) / w, d.get("SQ_INSTS_VMEM_RD", 0) / w,
                 d.get("SQ_INSTS_VMEM_WR", 0) / w, d.get("SQ_INSTS_SALU", 0) / w))
