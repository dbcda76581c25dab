#!/usr/bin/env python3
"""Builds the dragon stand-in's acceleration structures a few times (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
from pt_amd import hip, procedural
scene = procedural.dragon_standin()
for k in range(3):
    t0 = time.perf_counter()
    r = hip.Renderer(scene, 64, 64)
    r.ctx.sync()
    dt = time.perf_counter() - t0
    print("setup %.1f ms; build_ms" % (dt * 1e3), [round(float(b.info()["build_ms"]), 2) for b in r.blases], round(float(r.tlas.info()["build_ms"]), 2))
    r.close()
