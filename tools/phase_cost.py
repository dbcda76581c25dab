#!/usr/bin/env python3
"""What a phase iteration of the wave-scheduled traversal costs in vector instructions.

Run under `rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d DIR -- python3 tools/phase_cost.py run OUT.json`
with PT_HIP_LIB naming a library built with -DPT_SCHED_STATS: renders a list of (scene, size, spp, depth)
cases one after the other and stores every case's phase-iteration counts.  Then
`python3 tools/phase_cost.py fit OUT.json COUNTERS.csv` matches the k_intersect dispatches of the counter file to
the cases in order and solves  instructions = a*box + b*triangle + c*instance + d*outer + e*refill_rounds + f*waves
by least squares."""
import csv
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))

CASES = [("dragon", 1920, 1080, 4, 10), ("dragon", 1920, 1080, 8, 3), ("dragon", 1280, 720, 8, 2), ("cornell", 1920, 1080, 4, 10),
         ("cornell", 1280, 720, 8, 3), ("cube", 1920, 1080, 4, 10), ("cube", 1280, 720, 8, 2), ("helmet", 1920, 1080, 4, 10),
         ("helmet", 1280, 720, 8, 3), ("dragon", 960, 540, 16, 6), ("cornell", 960, 540, 16, 2), ("helmet", 960, 540, 16, 2)]


def run(out_path):
    from pt_amd import hip, procedural
    from pt_amd.scene import Scene
    os.environ.setdefault("PT_HIP_SCHED_MIN_RAYS", "0")      # every level in the scheduled form
    scenes = {}
    rows = []
    for name, w, h, spp, depth in CASES:
        if name not in scenes:
            scenes[name] = procedural.dragon_standin() if name == "dragon" else Scene.load_npz(os.path.join(ROOT, "tests", "golden", f"scene_{name}.npz"))
        r = hip.Renderer(scenes[name], w, h)
        lib = r.ctx.lib
        out = (C.c_ulonglong * 8)()
        lib.pt_debug_sched_stats(out, 1)
        r.reset_stats()
        r.render(spp=spp, max_depth=depth, frame=0)
        r.ctx.sync()
        lib.pt_debug_sched_stats(out, 1)
        st = r.stats()
        rows.append({"case": [name, w, h, spp, depth], "stats": [int(x) for x in out], "rays": int(st["rays"]),
                     "rays_per_depth": [int(x) for x in st["rays_per_depth"][:depth]], "intersect_launches": depth - 1})
        r.close()
    json.dump(rows, open(out_path, "w"), indent=1)


def fit(json_path, csv_path):
    rows = json.load(open(json_path))
    disp = {}
    for r in csv.DictReader(open(csv_path)):
        if r["Counter_Name"] != "SQ_INSTS_VALU":
            continue
        d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "v": 0.0, "grid": int(r.get("Grid_Size", 0) or 0)})
        d["v"] += float(r["Counter_Value"])
    inter = [disp[k] for k in sorted(disp) if "k_intersect" in disp[k]["name"]]
    at = 0
    A, y = [], []
    for row in rows:
        n = row["intersect_launches"]
        mine = inter[at:at + n]; at += n
        s = row["stats"]
        waves = sum(d["grid"] for d in mine) / 64.0
        A.append([s[0], s[1], s[2], s[6], s[7], waves])
        y.append(sum(d["v"] for d in mine))
    assert at == len(inter), (at, len(inter))
    A, y = np.array(A, float), np.array(y, float)
    x, *_ = np.linalg.lstsq(A, y, rcond=None)
    names = ["box step", "triangle step", "instance phase", "outer iteration", "refill round", "wave (start and end)"]
    for k, v in zip(names, x):
        print(f"{k:22s} {v:9.1f} vector instructions")
    pred = A @ x
    print("worst relative residual %.3f" % np.max(np.abs(pred - y) / y))
    for row, a, yy in zip(rows, A, y):
        rays2 = row["rays"] - row["rays_per_depth"][0]
        print(row["case"], "per 64 rays: instr %.0f box %.2f tri %.2f inst %.2f outer %.2f refill %.2f" % tuple(v / (rays2 / 64.0) for v in (yy, a[0], a[1], a[2], a[3], a[4])),
              " shares: " + " ".join("%s %.0f%%" % (n.split()[0], 100 * c * xx / yy) for n, c, xx in zip(names, a, x)))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        fit(sys.argv[2], sys.argv[3])
