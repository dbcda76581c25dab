#!/usr/bin/env python3
"""Merges the traffic_*.json entries a profile round left in a directory (tools/profile_round.sh on the GPU box,
merged back under gpurun_out/) into profiles/traffic.json.  usage: tools/merge_traffic.py <dir>"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(ROOT, "profiles", "traffic.json")
with open(path) as f:
    t = json.load(f)
for p in sorted(glob.glob(os.path.join(sys.argv[1], "traffic_*.json"))):
    e = json.load(open(p))
    key = e.pop("key")
    t[key] = e
    print("merged", key, "source_hash", e.get("source_hash"))
with open(path, "w") as f:
    json.dump(t, f, indent=1)
    f.write("\n")
