#!/usr/bin/env python3
"""HBM-side traffic of the traversal kernels (k_primary, k_intersect) from the FETCH_SIZE / WRITE_SIZE passes of
tools/pmc_passes.sh, per launch.  rocprofv3 reports both in KiB; on gfx950 FETCH_SIZE tallies the
128-byte requests of 16-byte-per-lane reads at 64 bytes, so it is doubled; WRITE_SIZE is exact.
Also the vector instructions per launch (SQ_INSTS_VALU), from which bench.py derives the issue rate.
usage: tools/pmc_traffic.py <pmc dir> <bench args recorded with the result...>  -> JSON on stdout"""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_hash

d = sys.argv[1]
tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "SQ_INSTS_VALU": 0.0}
launches = {"FETCH_SIZE": 0, "WRITE_SIZE": 0, "SQ_INSTS_VALU": 0}
for f in sorted(glob.glob(f"{d}/pass*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if ("k_intersect" in r["Kernel_Name"] or "k_primary" in r["Kernel_Name"]) and r["Counter_Name"] in tot:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            launches[r["Counter_Name"]] += 1
n = launches["FETCH_SIZE"]
assert n and n == launches["WRITE_SIZE"], launches
fetch = tot["FETCH_SIZE"] * 1024.0 * 2.0
write = tot["WRITE_SIZE"] * 1024.0
print(json.dumps({"kernel": "k_intersect + k_primary", "launches": n, "fetch_bytes_per_launch": fetch / n, "write_bytes_per_launch": write / n,
                  "bytes_per_launch": (fetch + write) / n, "corrections": "FETCH_SIZE KiB x 1024 x 2 (gfx950), WRITE_SIZE KiB x 1024",
                  # wave64 vector instructions the two kernels execute per launch (the SQ pass of the same workload)
                  "valu_insts_per_launch": tot["SQ_INSTS_VALU"] / launches["SQ_INSTS_VALU"] if launches["SQ_INSTS_VALU"] == n else None,
                  "source_hash": source_hash(),      # the kernel sources the counters belong to
                  "command": "bench.py " + " ".join(sys.argv[2:])}))
