#!/usr/bin/env python3
"""Per-launch counters of the tracing kernels (k_primary, k_intersect, k_shade) of one workload from the PMC passes
of tools/pmc_passes.sh, merged into profiles/traffic.json under the workload's key.
  SQ_INSTS_VALU            wave64 vector instructions (bench.py derives the issue rate from it)
  SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)   fraction of the lanes that take part in a vector instruction
  FETCH_SIZE, WRITE_SIZE   memory-side bytes: rocprofv3 reports both in KiB; on gfx950 FETCH_SIZE tallies the 128-byte
                           requests of 16-byte-per-lane reads at 64 bytes, so it is doubled; WRITE_SIZE is exact
usage: tools/pmc_traffic.py <pmc dir> --key "<scene> <W>x<H> <spp>spp depth<d>" [--write] -- <bench args recorded with the result...>"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import source_hash

d = sys.argv[1]
key = sys.argv[sys.argv.index("--key") + 1]
write = "--write" in sys.argv
cmd = sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else []
COUNTERS = ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU")
KERNELS = ("k_primary", "k_intersect", "k_shade")
tot = {k: {c: 0.0 for c in COUNTERS} for k in KERNELS}
cnt = {k: {c: 0 for c in COUNTERS} for k in KERNELS}
for f in sorted(glob.glob(f"{d}/pass*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        k = next((k for k in KERNELS if k + "<" in name or k + "(" in name or name.endswith(k)), None)
        if k is None or "k_primary_hits" in name or r["Counter_Name"] not in COUNTERS:
            continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
kernels = {}
for k in KERNELS:
    n = cnt[k]["SQ_INSTS_VALU"]
    if not n:
        continue
    assert all(cnt[k][c] == n for c in COUNTERS), (k, cnt[k])
    kernels[k] = {"launches": n, "valu_insts_per_launch": tot[k]["SQ_INSTS_VALU"] / n,
                  "lane_utilisation": tot[k]["SQ_THREAD_CYCLES_VALU"] / (64.0 * tot[k]["SQ_ACTIVE_INST_VALU"]),
                  "fetch_bytes_per_launch": tot[k]["FETCH_SIZE"] * 1024.0 * 2.0 / n, "write_bytes_per_launch": tot[k]["WRITE_SIZE"] * 1024.0 / n}
trav = [kernels[k] for k in ("k_primary", "k_intersect") if k in kernels]
n = sum(x["launches"] for x in trav)
entry = {"key": key, "kernel": "k_intersect + k_primary", "launches": n,
         "fetch_bytes_per_launch": sum(x["fetch_bytes_per_launch"] * x["launches"] for x in trav) / n,
         "write_bytes_per_launch": sum(x["write_bytes_per_launch"] * x["launches"] for x in trav) / n,
         "valu_insts_per_launch": sum(x["valu_insts_per_launch"] * x["launches"] for x in trav) / n,
         "lane_utilisation": sum(tot[k]["SQ_THREAD_CYCLES_VALU"] for k in ("k_primary", "k_intersect")) /
                             (64.0 * sum(tot[k]["SQ_ACTIVE_INST_VALU"] for k in ("k_primary", "k_intersect"))),
         "kernels": kernels,
         "corrections": "FETCH_SIZE KiB x 1024 x 2 (gfx950), WRITE_SIZE KiB x 1024; lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)",
         "source_hash": source_hash(),      # the kernel sources the counters belong to
         "command": "bench.py " + " ".join(cmd)}
entry["bytes_per_launch"] = entry["fetch_bytes_per_launch"] + entry["write_bytes_per_launch"]
print(json.dumps(entry))
if write:
    path = os.path.join(ROOT, "profiles", "traffic.json")
    with open(path) as f:
        t = json.load(f)
    t[key] = {k: v for k, v in entry.items() if k != "key"}
    with open(path, "w") as f:
        json.dump(t, f, indent=1)
        f.write("\n")
