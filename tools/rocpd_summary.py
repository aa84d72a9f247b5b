#!/usr/bin/env python3
"""Dump the per-kernel statistics (rocprofv3 --kernel-trace --stats) from a rocpd .db or *_kernel_stats.csv
into a small text table that can be committed under profiles/."""
import csv
import sqlite3
import sys
from pathlib import Path


def from_db(path, top=15):
    db = sqlite3.connect(path)
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    return [(n, int(c), float(t), float(a), float(p)) for n, c, t, a, p in rows][:top]


def from_csv(path, top=15):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3,
                        float(r["Percentage"])))
    return out[:top]


def demangle(name):
    """rocprofv3 leaves names with _Float16 parameters mangled (its demangler does not know DF16_): c++filt does, as Dh"""
    if not name.startswith("_Z"):
        return name
    import subprocess
    try:
        out = subprocess.run(["c++filt", name.replace("DF16_", "Dh")], capture_output=True, text=True).stdout.strip()
        return out.replace("__fp16", "_Float16") or name
    except OSError:
        return name


def main():
    path = Path(sys.argv[1])
    rows = from_db(path) if path.suffix == ".db" else from_csv(path)
    rows = [(demangle(n), c, t, a, p) for n, c, t, a, p in rows]
    print("%-100s %6s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "%"))
    for n, c, t, a, p in rows:
        print("%-100s %6d %14.1f %12.1f %7.2f" % (n[:100], c, t, a, p))


if __name__ == "__main__":
    main()
