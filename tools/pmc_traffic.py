#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/gpu_pmc_full.sh:
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (KB counters; gfx950 FETCH_SIZE reports half the bytes of wide
coalesced reads, MI355X_MICROARCH.md), mean over the dispatches of each kernel."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

root, out = sys.argv[1], sys.argv[2]
vals = {"FETCH_SIZE": defaultdict(list), "WRITE_SIZE": defaultdict(list)}
for c in vals:
    for f in glob.glob("%s/%s/**/*counter_collection.csv" % (root, c), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != c:
                    continue
                m = re.search(r"(als_(?:cgq|ne)_kernel<[^>]*>)", r.get("Kernel_Name", ""))
                if m:
                    vals[c][m.group(1)].append(float(r["Counter_Value"]))
res = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KB per dispatch, mean over the dispatches of "
               "`python bench.py --steps 1 --warmup 0 --no-cpu-baseline`); hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per "
               "MI355X_MICROARCH.md (gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads; Infinity-Cache hits "
               "are counted)", "detail": {}}
for k in sorted(vals["FETCH_SIZE"]):
    f = sum(vals["FETCH_SIZE"][k]) / len(vals["FETCH_SIZE"][k])
    w = sum(vals["WRITE_SIZE"][k]) / max(1, len(vals["WRITE_SIZE"][k])) if vals["WRITE_SIZE"][k] else 0.0
    res[k] = (2 * f + w) * 1024
    res["detail"][k] = {"fetch_kb_per_launch": f, "write_kb_per_launch": w, "hbm_bytes_per_launch": res[k],
                        "dispatches": len(vals["FETCH_SIZE"][k])}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k.startswith("als_")}, indent=1))
