#!/usr/bin/env python3
"""Throughput of the dragon 1080p workload against samples per pt_trace call and chunk size."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
from pt_amd import hip, procedural

scene = procedural.dragon_standin()
W, H = 1920, 1080
r = hip.Renderer(scene, W, H)
for spp, max_paths in ((16, 1 << 25), (16, 1 << 23), (32, 1 << 26), (64, 1 << 27), (64, 1 << 25)):
    r.ctx.set_max_paths(max_paths)
    r.render(spp=spp, max_depth=10, frame=0)
    r.ctx.sync()
    s0 = r.stats()
    t0 = time.perf_counter()
    n = max(1, 64 // spp)
    for f in range(n):
        r.render(spp=spp, max_depth=10, frame=1 + f)
    r.ctx.sync()
    dt = time.perf_counter() - t0
    s1 = r.stats()
    rays = int(s1["rays"]) - int(s0["rays"])
    print(f"spp/call {spp:3d} max_paths 2^{max_paths.bit_length() - 1}: {rays / dt / 1e9:.3f} Grays/s, {dt / n * 1e3 * 16 / spp:.2f} ms per 16 spp")
r.close()
