#!/usr/bin/env python3
"""Builds the dragon stand-in's acceleration structures three times (the first build also sizes the arena).
Under `rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/build_trace.py`, then
`python3 tools/build_trace.py analyse DIR/.../*_kernel_trace.csv` prints the last build's timeline summary."""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))

def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return name.split("(")[0].split("<")[0][-30:]


if len(sys.argv) > 2 and sys.argv[1] == "analyse":
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
    # the builds are separated by the k_tri_records / idle gaps; take the kernels after the last k_blas_maxabs pair start
    starts = [i for i, r in enumerate(rows) if "k_blas_maxabs" in r["Kernel_Name"]]
    n_blas = 2
    first = starts[-n_blas]
    sel = rows[first:]
    t0, t1 = int(sel[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in sel)
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel)
    print(f"last build: {len(sel)} kernels, span {(t1 - t0) / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us")
    gaps = []
    for a, b in zip(sel, sel[1:]):
        gaps.append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"]), short(a["Kernel_Name"]), short(b["Kernel_Name"])))
    big = sorted(gaps, reverse=True)[:12]
    print("largest gaps (us):", [(round(g / 1e3, 1), a, b) for g, a, b in big])
    import collections
    per = collections.defaultdict(lambda: [0, 0])
    for r in sel:
        k = short(r["Kernel_Name"])
        per[k][0] += 1; per[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"  {k:28s} {c:4d} x {t / c / 1e3:7.2f} us = {t / 1e3:8.1f} us")
    gsum = sum(g for g, _, _ in gaps)
    small = sum(g for g, _, _ in gaps if g < 20000)
    print(f"gaps: total {gsum / 1e3:.1f} us, of which below 20 us each {small / 1e3:.1f} us in {sum(1 for g, _, _ in gaps if g < 20000)} gaps")
    sys.exit(0)

from pt_amd import hip, procedural
scene = procedural.dragon_standin()
for _ in range(3):
    r = hip.Renderer(scene, 64, 64)
    print("build ms", sum(float(b.info()["build_ms"]) for b in r.blases), float(r.tlas.info()["build_ms"]))
    r.close()
