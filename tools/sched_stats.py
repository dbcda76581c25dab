#!/usr/bin/env python3
"""Phase statistics of the wave-scheduled traversal (needs a library built with -DPT_SCHED_STATS,
selected through PT_HIP_LIB).  Prints wave-level iterations and lane utilisation per phase."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
from pt_amd import hip, procedural

if len(sys.argv) > 1 and sys.argv[1] == "cornell":
    from pt_amd.scene import Scene
    scene = Scene.load_npz(os.path.join(ROOT, "tests", "golden", "scene_cornell.npz"))
else:
    scene = procedural.dragon_standin()
W, H, SPP = 1920, 1080, int(os.environ.get("SPP", "4"))
r = hip.Renderer(scene, W, H)
lib = r.ctx.lib
out = (C.c_ulonglong * 8)()
print("rc", lib.pt_debug_sched_stats(out, 1))
r.render(spp=SPP, max_depth=10, frame=0)
r.ctx.sync() if hasattr(r.ctx, "sync") else None
print("rc", lib.pt_debug_sched_stats(out, 1), lib.pt_last_error())
st = r.stats()
# bounce 0 runs in k_primary (plain schedule, not counted): the phases below serve the later bounces' rays only
rays = int(st["rays"]) - int(st["rays_per_depth"][0])
v = [int(x) for x in out]
print("rays of bounces 1+", rays, "of", int(st["rays"]))
for k, name in enumerate(("node", "triangle", "enter")):
    it, act = v[k], v[3 + k]
    print(f"{name:9s} wave iterations {it:12d}  per 64 rays {it / (rays / 64):8.2f}  lanes active {act / max(it, 1):6.2f}  lane-steps per ray {act / rays:7.2f}")
print("outer iterations per 64 rays", v[6] / (rays / 64), " refill rounds per 64 rays", v[7] / (rays / 64))
r.close()
