#!/usr/bin/env python3
"""Node visits and triangle tests per ray of the 4-wide layout (children sorted by distance, leaves of up to
four triangles) and the 8-wide layout (octant order, one triangle per leaf child) on the same rays, simulated
on the host with the builders' own per-node routines (make -C experimental).  usage: experimental/sim_layouts.py [mesh 0|1] [rays]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
from pt_amd import abi, procedural

lib = C.CDLL(os.path.join(ROOT, "experimental", "_build", "libpt_layout_sim.so"))
lib.pt_sah_build_host_c.restype = C.c_uint32
lib.pt_wide_collapse_host.restype = C.c_uint32
lib.pt_wide8_collapse_host.restype = C.c_uint32
P = lambda a: a.ctypes.data_as(C.c_void_p)

mesh_id = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
scene = procedural.dragon_standin()
model = scene.models[0]
mesh = model.meshes[mesh_id]
idx = model.indices[mesh.first_index:mesh.first_index + mesh.index_count].reshape(-1, 3)
pos = model.vertices["pos"][idx]                      # n x 3 x 3
n = len(pos)
delta = np.float32(np.abs(pos).max() * 2.0 ** -15)
boxes = np.concatenate([pos.min(1) - delta, pos.max(1) + delta], axis=1).astype(np.float32)
root_lo, root_hi = boxes[:, :3].min(0).copy(), boxes[:, 3:].max(0).copy()


def sah(max_leaf):
    nodes = np.zeros(n, dtype=abi.BVH_NODE); order = np.zeros(n, np.uint32)
    cnt = lib.pt_sah_build_host_c(P(boxes), n, max_leaf, P(root_lo), P(root_hi), P(nodes), P(order))
    return nodes[:cnt].copy(), order


rng = np.random.default_rng(7)
# rays leaving the surface: origin = a point of a random triangle pushed off along its normal, direction in the
# hemisphere of the normal; and rays from outside aimed into the box
tri = rng.integers(0, n, n_rays)
b = rng.dirichlet([1, 1, 1], n_rays).astype(np.float32)
p0 = (pos[tri] * b[:, :, None]).sum(1)
nrm = np.cross(pos[tri, 1] - pos[tri, 0], pos[tri, 2] - pos[tri, 0]); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True) + 1e-30
d = rng.normal(size=(n_rays, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
d = np.where((d * nrm).sum(1, keepdims=True) < 0, -d, d)
o_surf = (p0 + 1e-3 * nrm).astype(np.float32); d_surf = d.astype(np.float32)
ext = root_hi - root_lo
o_out = (root_lo - ext + rng.uniform(0, 3, size=(n_rays, 3)) * ext).astype(np.float32)
tgt = root_lo + rng.uniform(0, 1, size=(n_rays, 3)) * ext
d_out = (tgt - o_out); d_out = (d_out / np.linalg.norm(d_out, axis=1, keepdims=True)).astype(np.float32)

# 4-wide
nodes4, order4 = sah(4)
w4 = np.zeros(len(nodes4), dtype=abi.WIDE_NODE); depth = C.c_uint32()
c4 = lib.pt_wide_collapse_host(P(nodes4), len(nodes4), 0, P(w4), len(w4), C.byref(depth))
tris4 = np.ascontiguousarray(pos[order4].reshape(n, 9), np.float32)
# 8-wide
nodes1, order1 = sah(1)
w8 = np.zeros(len(nodes1) * 80, dtype=np.uint8); ord8 = np.zeros(n, np.uint32); depth8 = C.c_uint32()
c8 = lib.pt_wide8_collapse_host(P(nodes1), len(nodes1), 0, P(w8), len(nodes1), P(ord8), C.byref(depth8))
tris8 = np.ascontiguousarray(pos[order1[ord8]].reshape(n, 9), np.float32)
print(f"mesh {mesh_id}: {n} triangles; binary nodes {len(nodes4)} (leaf<=4) / {len(nodes1)} (leaf 1); 4-wide {c4} nodes depth {depth.value} ({c4 * 64 / 1e6:.2f} MB); "
      f"8-wide {c8} nodes depth {depth8.value} ({c8 * 80 / 1e6:.2f} MB)")
for name, o, dd in (("surface", o_surf, d_surf), ("outside", o_out, d_out)):
    s4 = np.zeros(24, np.uint64); s8 = np.zeros(3, np.uint64)
    t4 = np.zeros(n_rays, np.float32); t8 = np.zeros(n_rays, np.float32)
    lib.pt_wide4_simulate_host(P(w4), P(tris4), n_rays, P(o), P(dd), C.c_float(1e-3), C.c_float(1e4), P(s4), P(t4), 0)
    for mode in (1, 2, 3, 5):
        sm = np.zeros(24, np.uint64); tm = np.zeros(n_rays, np.float32)
        lib.pt_wide4_simulate_host(P(w4), P(tris4), n_rays, P(o), P(dd), C.c_float(1e-3), C.c_float(1e4), P(sm), P(tm), mode)
        assert np.array_equal(tm, t4)
        print(f"    4-wide order mode {mode}: {sm[0] / n_rays:6.2f} nodes {sm[1] / n_rays:5.2f} tris (x{sm[0] / s4[0]:.3f} nodes, x{sm[1] / s4[1]:.3f} tris)")
    lib.pt_wide8_simulate_host(P(w8), P(tris8), n_rays, P(o), P(dd), C.c_float(1e-3), C.c_float(1e4), P(s8), P(t8))
    assert np.array_equal(t4, t8), (np.sum(t4 != t8))
    print("    4-wide: share of node steps leaving >= k entries on the stack:", " ".join(f"{k}:{s4[3 + k] / s4[0]:.3f}" for k in (4, 6, 8, 10, 12, 14, 16)))
    print(f"{name:8s} hit {np.mean(t4 < 1e4):.2f} | 4-wide: {s4[0] / n_rays:6.2f} nodes {s4[1] / n_rays:5.2f} tris, stack {s4[2]} | "
          f"8-wide: {s8[0] / n_rays:6.2f} nodes {s8[1] / n_rays:5.2f} tris, stack {s8[2]} | node ratio {s8[0] / s4[0]:.3f}")
