#!/usr/bin/env python3
"""gpurun_out/wrmf_core_errors.jsonl (written by tests/test_wrmf_core.py on the GPU box) -> tests/golden/
wrmf_core_tolerances.json: per cell of the reference's test grid the achieved error of the device path against the
fp64 oracle (max over components / user embeddings / user-loss sequence, max over the recorded runs) and the bound the
test asserts: 3x the achieved error rounded up to one significant digit, never below the north star's 1e-4."""
import json
import math
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
src = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "gpurun_out" / "wrmf_core_errors.jsonl"
cells = {}
for line in src.read_text().splitlines():
    r = json.loads(line)
    c = cells.setdefault(r["cell"], {"components": 0.0, "user_emb": 0.0, "loss": 0.0})
    for key in ("components", "user_emb", "loss"):
        c[key] = max(c[key], r[key])


def round_up(v):
    e = math.floor(math.log10(v))
    return math.ceil(v / 10 ** e) * 10 ** e


for c in cells.values():
    worst = max(c.values())
    c["achieved"] = worst
    c["bound"] = max(1e-4, round_up(3.0 * worst)) if worst > 0 else 1e-4
out = {"what": "tests/test_wrmf_core.py: per-cell achieved error vs the fp64 oracle and the asserted bound "
               "(cell = feedback|solver|lambda|with_user_item_bias|precision)",
       "cells": dict(sorted(cells.items()))}
(ROOT / "tests" / "golden" / "wrmf_core_tolerances.json").write_text(json.dumps(out, indent=1) + "\n")
print("%d cells, %d above 1e-4" % (len(cells), sum(c["bound"] > 1e-4 for c in cells.values())))
