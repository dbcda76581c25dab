#!/usr/bin/env python3
"""Throughput of the dragon 1080p workload against the BLAS leaf size (the closest hit does not depend on it)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
from pt_amd import hip, procedural

scene = procedural.dragon_standin()
W, H, SPP = 1920, 1080, 64
for max_leaf in [int(x) for x in os.environ.get("LEAVES", "1,2,3,4,6,8").split(",")]:
    r = hip.Renderer(scene, W, H, max_leaf=max_leaf)
    for f in range(3):
        r.render(spp=SPP, max_depth=10, frame=f)
    r.ctx.sync()
    s0 = r.stats()
    t0 = time.perf_counter()
    for f in range(6):
        r.render(spp=SPP, max_depth=10, frame=10 + f)
    r.ctx.sync()
    dt = time.perf_counter() - t0
    s1 = r.stats()
    rays = int(s1["rays"]) - int(s0["rays"])
    print(f"max_leaf {max_leaf}: {rays / dt / 1e9:.3f} Grays/s, wide nodes {sum(len(b.download_wide()[0]) for b in r.blases)}")
    r.close()
