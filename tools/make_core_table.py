#!/usr/bin/env python3
"""gpurun_out/wrmf_core_errors.jsonl (written by tests/test_wrmf_core.py on the GPU box) -> a two-column report,
profiles/r04/wrmf_core_parity_table.md: per cell of the reference's test grid the error of the DEVICE fit and of the
ORACLE-IN-FLOAT fit, both against the fp64 oracle, and the bound the test derived from the latter.  A report: the test
computes its bound itself and reads nothing back."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
src = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "gpurun_out" / "wrmf_core_errors.jsonl"
dst = Path(sys.argv[2]) if len(sys.argv) > 2 else ROOT / "profiles" / "r04" / "wrmf_core_parity_table.md"
cells = {}
for line in src.read_text().splitlines():
    r = json.loads(line)
    if "device" in r:
        cells[r["cell"]] = r          # the last run of a cell wins
rows = ["# tests/test_wrmf_core.py -- device vs fp32 oracle, both against the fp64 oracle", "",
        "54 fits of the reference's grid (tests/testthat/test-wrmf.R:9-90; movielens100k rows 1:900, 5 iterations).",
        "err = max(relative Frobenius error of the item factors, of the user embeddings, relative error of the user-side",
        "loss sequence).  precision = double cells: the device computes in double (wrmf_f64.hip), asserted `device <= 1e-4` flat.",
        "precision = float cells, asserted: `device <= max(1e-4, 2 x fp32 oracle)`; fp32 oracle = the largest distance from the fp64 fit over",
        "the number of fp32-oracle fits in the `fits` column (1 where the first is below 3e-5, else 5: the given initial factors and",
        "four one-ulp-scale perturbations of them -- in the cells above 1e-4 the fp32 fit is a noisy trajectory).", "",
        "| cell (feedback, solver, lambda, biases, precision) | rank | device | fp32 oracle | fits | bound | device / fp32 oracle |",
        "|---|---|---|---|---|---|---|"]
n_above = 0
for cell, r in sorted(cells.items()):
    d = max(r["device"].values())
    n_above += d > 1e-4
    if not r.get("fp32_oracle"):      # a double cell: no yardstick
        rows.append("| %s | %d | %.2e | -- | 0 | %.1e | -- |" % (cell.replace("|", ", "), r["rank"], d, r["bound"]))
        continue
    y = max(r["fp32_oracle"].values())
    rows.append("| %s | %d | %.2e | %.2e | %d | %.1e | %.2f |" % (cell.replace("|", ", "), r["rank"], d, y, r.get("fp32_fits", 1), r["bound"], d / max(y, 1e-300)))
rows += ["", "%d cells; %d with a device error above 1e-4 (all of them precision = float cells in which the fp32 oracle is above 1e-4 too)." % (len(cells), n_above)]
dst.parent.mkdir(parents=True, exist_ok=True)
dst.write_text("\n".join(rows) + "\n")
print("%d cells -> %s" % (len(cells), dst))
