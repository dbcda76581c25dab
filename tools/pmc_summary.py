#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSVs written by tools/pmc_passes.sh: per kernel (and per launch of
the intersect / shade kernels of the last traced frame) the counter values.
usage: tools/pmc_summary.py <dir> [--per-launch] [--all]"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
per_launch = "--per-launch" in sys.argv
vals = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> [values in dispatch order]
for f in sorted(glob.glob(f"{d}/pass*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        short = "k_" + k.split("k_", 1)[1].split("(")[0].split("<")[0] if "k_" in k else k[:40]
        vals[short][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for k in sorted(vals):
    if k not in ("k_intersect", "k_primary", "k_shade", "k_accumulate") and "--all" not in sys.argv:
        continue                      # the one-time builder kernels only with --all
    print(f"== {k}")
    for c in sorted(vals[k]):
        v = [x for _, x in sorted(vals[k][c])]
        line = f"   {c:34s} n={len(v):3d} sum={sum(v):.4g}"
        if per_launch and k in ("k_intersect", "k_shade"):
            line += "  last10=" + " ".join(f"{x:.3g}" for x in v[-10:])
        print(line)
