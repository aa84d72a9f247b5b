#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/gpu_pmc_full.sh:
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (KB counters; gfx950 FETCH_SIZE reports half the bytes of wide
coalesced reads, MI355X_MICROARCH.md), mean over the dispatches of each kernel.  The table records the WORKLOAD it was
collected on (from the bench line the profiled run printed); bench.py uses it only for that workload.

    python tools/pmc_traffic.py gpurun_out/<tag> profiles/pmc_traffic.json"""
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
                kn = r.get("Kernel_Name", "")
                m = re.search(r"((?:als|gramian|top_product)\w*<[^()]*>)\s*\(", kn) or re.match(r"(rsparse_hip_\w+)", kn)   # (extern "C": the bare name)
                if m:
                    vals[c][m.group(1)].append(float(r["Counter_Value"]))
workload = None
for line in open("%s/FETCH_SIZE.log" % root, errors="replace"):
    if line.startswith("{") and '"metric"' in line:
        cfg = json.loads(line)["config"]
        workload = {"users": cfg["n_users"], "items": cfg["n_items"], "nnz": cfg["nnz"], "rank": cfg["rank"],
                    "feedback": cfg["feedback"], "solver": {"conjugate_gradient": "cg"}.get(cfg["solver"], cfg["solver"]),
                    "cg_steps": cfg["cg_steps"], "n_gpus": 1}
res = {"how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel-trace only; KB per dispatch, mean over the "
              "dispatches of `python bench.py --steps 1 --warmup 0 --no-cpu-baseline`); hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
              "per MI355X_MICROARCH.md (gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads; Infinity-Cache hits "
              "are counted)",
       "workload": workload, "kernels": {}}
for k in sorted(vals["FETCH_SIZE"]):
    f = sum(vals["FETCH_SIZE"][k]) / len(vals["FETCH_SIZE"][k])
    w = sum(vals["WRITE_SIZE"][k]) / max(1, len(vals["WRITE_SIZE"][k])) if vals["WRITE_SIZE"][k] else 0.0
    res["kernels"][k] = {"fetch_kb_per_launch": f, "write_kb_per_launch": w, "hbm_bytes_per_launch": (2 * f + w) * 1024,
                         "dispatches": len(vals["FETCH_SIZE"][k])}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({"workload": workload, **{k: v["hbm_bytes_per_launch"] for k, v in res["kernels"].items() if k.startswith("als_") or k.startswith("rsparse_hip_")}}, indent=1))
