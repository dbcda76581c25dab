import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/path-tracer_amd")
import numpy as np
import bench
from pt_amd import hip
def area(lo, hi):
    d = np.maximum(hi - lo, 0); return d[0]*d[1] + d[1]*d[2] + d[2]*d[0]
def cost(nodes):
    lo = np.minimum(nodes["lo0"][0], nodes["lo1"][0]); hi = np.maximum(nodes["hi0"][0], nodes["hi1"][0])
    root = area(lo, hi)
    c = 0.0
    for nd in nodes:
        c += (area(nd["lo0"], nd["hi0"]) + area(nd["lo1"], nd["hi1"])) / root
    return c
for name in ("cornell", "helmet", "composite", "dragon"):
    r = hip.Renderer(bench.build_scene(name), 16, 16)
    m = r.tlas.download()[0]; s = r.tlas.download_traversal_tree()[0]
    print(name, "instances", r.tlas.n, "morton cost %.3f" % cost(m), "sweep cost %.3f" % cost(s), "wide nodes", len(r.tlas.download_wide()[0]))
    r.close()
