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
            short = name.split("(")[0].replace("void rsparse_hip::(anonymous namespace)::", "")
            agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print("   %-28s n=%3d mean=%.4g" % (c, len(v), sum(v) / len(v)))
