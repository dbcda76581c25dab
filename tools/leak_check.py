#!/usr/bin/env python3
"""Creates and destroys renderers repeatedly and checks that free device memory does not shrink (run on the GPU box)."""
import os, sys
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import torch
from pt_amd import hip, procedural
from pt_amd.scene import Scene
scene = procedural.dragon_standin()
corn = Scene.load_npz(os.path.join(ROOT, "tests", "golden", "scene_cornell.npz"))
free = []
for k in range(12):
    for sc in (scene, corn):
        r = hip.Renderer(sc, 256, 256)
        r.render(spp=8, frame=0); r.render(spp=8, frame=1)
        r.ctx.sync()
        r.close()
    torch.cuda.synchronize()
    free.append(torch.cuda.mem_get_info()[0] >> 20)
print("free MiB after each round:", free)
assert max(free[2:]) - min(free[2:]) < 64, free
print("no growth")
