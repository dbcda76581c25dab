#!/usr/bin/env python3
"""Prints the rows of BASELINE.md's table of measurements from the committed bench lines: tools/baseline_table.py [prefix]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prefix = sys.argv[1] if len(sys.argv) > 1 else "r03_v4"


def line(name):
    with open(os.path.join(ROOT, "profiles", f"{prefix}_{name}.json")) as f:
        return json.loads([l for l in f if l.startswith('{"metric')][-1])


def cpu_row(label, j):
    c = j["cpu_baseline"]
    return f"| {label} | CPU oracle (`oracle/`, kind `port`) | {c['cores']} threads (the job's share of the host) | {c['value'] / 1e6:.1f} M | {c['paths_per_s'] / 1e6:.1f} M | — | n/a |"


def gpu_row(label, j):
    r = j["roofline"]
    sm = r.get("survey_model")
    b = f"{sm['bytes_per_ray']:.0f} B (model: {sm['modelled_gbs'] / 8000:.2f} × 8 TB/s — not a bound, DESIGN §4)" if sm else "—"
    h = f"{r['hbm']['frac']:.2f} measured (VALU issue {r['frac']:.2f}, lanes {r['lane_utilisation']:.2f}, useful {r['useful_frac']:.2f})" if r.get("frac") else "not collected"
    return f"| {label} | HIP | 1 | {j['value'] / 1e9:.2f} G | {j['paths_per_s'] / 1e9:.2f} G | {b} | {h} |"


cfgs = [("cornell 512², 16 spp (cfg 1)", "cfg1_cornell_512"), ("cornell 1080p, 16 spp", "cfg2_cornell_1080p"), ("cube 1080p, 16 spp", "cfg2_cube_1080p"),
        ("helmet 1080p, 16 spp", "cfg3_helmet_1080p"), ("dragon stand-in 1080p, 16 spp", "bench"), ("composite 3840×2160, 2 spp", "cfg5_composite_2160p")]
for label, name in cfgs:
    print(cpu_row(label, line(name)))
for label, name in [("cornell 512², 16 spp, one step (cfg 1)", "cfg1_cornell_512"), ("cornell 1080p, 256 spp × 4 steps (cfg 2)", "cfg2_cornell_1080p"),
                    ("cube 1080p, 256 spp × 4 steps (cfg 2)", "cfg2_cube_1080p"), ("helmet 1080p, 1024 spp, native texture footprint (cfg 3)", "cfg3_helmet_1080p"),
                    ("dragon stand-in 1080p, 256 spp × 8 steps (the headline line)", "bench"),
                    ("dragon stand-in 1080p, 4096 spp as one image (cfg 4, strong-scaling mode at N = 1)", "cfg4_dragon_1080p_4096spp"),
                    ("composite 3840×2160, 256 spp (cfg 5 at reduced spp; 64k spp on 8 GPUs is the driver's run)", "cfg5_composite_2160p")]:
    print(gpu_row(label, line(name)))
