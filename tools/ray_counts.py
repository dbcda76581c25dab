#!/usr/bin/env python3
"""Rays per bounce of one frame of the bench workload (run on the GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
from pt_amd import hip, procedural

if len(sys.argv) > 1 and sys.argv[1] == "cornell":
    from pt_amd.scene import Scene
    scene = Scene.load_npz(os.path.join(ROOT, "tests", "golden", "scene_cornell.npz"))
else:
    scene = procedural.dragon_standin()
r = hip.Renderer(scene, 1920, 1080)
r.render(spp=16, max_depth=10, frame=0)
r.ctx.sync()
st = r.stats()
per = [int(x) for x in st["rays_per_depth"][:10]]
print("rays per bounce", per)
print("survival", [round(per[i + 1] / per[i], 3) for i in range(9)])
r.close()
