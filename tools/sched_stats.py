#!/usr/bin/env python3
"""Phase statistics of the wave-scheduled traversal (needs a library built with -DPT_SCHED_STATS,
selected through PT_HIP_LIB: tools/build_variant.sh stats -DPT_SCHED_STATS).  Prints wave-level iterations and lane
utilisation per phase for the rays of bounces 1+ and, beside them, what the CPU port counts on its canonical binary
tree for a reduced image of the same scene and camera (box and triangle tests per ray over all bounces).
usage: tools/sched_stats.py [dragon|cornell|cube|helmet|composite] [width height]     (SPP=samples per step, default 4)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import bench
from pt_amd import hip

name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
SPP = int(os.environ.get("SPP", "4"))
scene = bench.build_scene(name)
cam = bench.scene_camera(name, W, H)
r = hip.Renderer(scene, W, H, camera=cam)
lib = r.ctx.lib
out = (C.c_ulonglong * 8)()
print("scene", name, W, H, "spp", SPP, "rc", lib.pt_debug_sched_stats(out, 1))
r.ctx.set_frames_in_flight(1)
r.ctx.set_max_paths(min(W * H * SPP, (1 << 31) - 64))
r.render(spp=SPP, max_depth=10, frame=0)
r.ctx.sync()
print("rc", lib.pt_debug_sched_stats(out, 1), lib.pt_last_error())
st = r.stats()
# bounce 0 runs in k_primary (plain schedule, not counted): the phases below serve the later bounces' rays only
rays = int(st["rays"]) - int(st["rays_per_depth"][0])
v = [int(x) for x in out]
print("rays of bounces 1+", rays, "of", int(st["rays"]), " per depth", [int(x) for x in st["rays_per_depth"][:10]])
for k, nm in enumerate(("node", "triangle", "enter")):
    it, act = v[k], v[3 + k]
    print(f"{nm:9s} wave iterations {it:12d}  per 64 rays {it / (rays / 64):8.2f}  lanes active {act / max(it, 1):6.2f}  lane-steps per ray {act / rays:7.2f}")
print("outer iterations per 64 rays", v[6] / (rays / 64), " refill rounds per 64 rays", v[7] / (rays / 64))
r.close()
if os.environ.get("CPU", "1") == "1":
    w2, h2 = max(W // 8, 16), max(H // 8, 16)
    c = bench.cpu_baseline(scene, bench.scene_camera(name, w2, h2), w2, h2, 2, 10)
    print(f"CPU port, {w2}x{h2}, 2 spp, all bounces: box tests per ray {c['box_tests_per_ray']:.2f}  triangle tests per ray {c['tri_tests_per_ray']:.2f}  hit fraction {c['hit_fraction']:.3f}")
