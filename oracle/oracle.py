"""ctypes wrapper of the CPU oracle (oracle/_build/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by the
cpu_baseline leg of bench.py -- never by the product package.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
sys.path.insert(0, os.path.join(os.path.dirname(_HERE), "path-tracer_amd"))
from pt_amd import abi  # noqa: E402


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def _load():
    if not os.path.exists(_LIB_PATH):
        build()
    lib = C.CDLL(_LIB_PATH)
    u32, u64, f, vp, i = C.c_uint32, C.c_uint64, C.c_float, C.c_void_p, C.c_int
    sig = {
        "or_tea": (u32, [u32, u32]),
        "or_lcg": (u32, [vp]),
        "or_random": (f, [vp]),
        "or_sincos": (None, [f, vp, vp]),
        "or_create_coordinate_system": (None, [vp, vp, vp]),
        "or_sampling_hemisphere": (None, [vp, vp, vp, vp, vp]),
        "or_primary_ray": (None, [vp, u32, u32, u32, u32, vp, vp]),
        "or_invert_transform": (None, [vp, vp]),
        "or_blas_build": (vp, [vp, u32, u32, vp, u32, u32]),
        "or_blas_destroy": (None, [vp]),
        "or_blas_node_count": (u32, [vp]),
        "or_blas_nodes": (vp, [vp]),
        "or_blas_leaf_prims": (vp, [vp]),
        "or_blas_sorted_keys": (vp, [vp]),
        "or_blas_root_box": (None, [vp, vp, vp]),
        "or_blas_delta": (f, [vp]),
        "or_tlas_build": (vp, [vp, u32]),
        "or_tlas_destroy": (None, [vp]),
        "or_tlas_node_count": (u32, [vp]),
        "or_tlas_nodes": (vp, [vp]),
        "or_tlas_leaf_instances": (vp, [vp]),
        "or_tlas_instance_box": (None, [vp, u32, vp, vp]),
        "or_tlas_instance_w2o": (None, [vp, u32, vp]),
        "or_scene_create": (vp, [vp, vp, u32, vp, u32, vp, u32, vp, u32]),
        "or_scene_destroy": (None, [vp]),
        "or_intersect": (None, [vp, u32, vp, vp, f, f, i, vp, vp, vp]),
        "or_intersect_counts": (None, [vp, u32, vp, vp, f, f, vp, vp]),
        "or_render_frame": (None, [vp, vp, u32, u32, u32, u32, u32, u32, u32, u32, u32, i, i, vp, vp]),
        "or_trace_path": (u32, [vp, vp, u32, u32, u32, u32, u32, u32, vp, vp]),
        "or_accumulate": (None, [i, u32, u32, vp, vp]),
        "or_resolve": (None, [u32, vp, vp]),
        "or_texture_bilinear_channel": (f, [vp, f, f, i]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()

STATS = np.dtype([("rays", "<u8"), ("box_tests", "<u8"), ("tri_tests", "<u8"), ("hits", "<u8"), ("tex_hits", "<u8"), ("paths", "<u8")])
BOUNCE = np.dtype([("hit", abi.HIT), ("t", "<f4"), ("origin", "<f4", 3), ("dir", "<f4", 3),
                   ("hit_value", "<f4", 3), ("weight", "<f4", 3), ("seed_after", "<u4")])


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def tea(a, b):
    return int(lib.or_tea(a & 0xFFFFFFFF, b & 0xFFFFFFFF))


def lcg_sequence(state, n):
    s = np.array([state], dtype=np.uint32)
    out, states, floats = [], [], []
    for _ in range(n):
        out.append(int(lib.or_lcg(_p(s))))
        states.append(int(s[0]))
    return states, out


def random_sequence(state, n):
    s = np.array([state], dtype=np.uint32)
    return [float(lib.or_random(_p(s))) for _ in range(n)]


def sincos(x):
    s = np.zeros(1, np.float32)
    c = np.zeros(1, np.float32)
    lib.or_sincos(np.float32(x), _p(s), _p(c))
    return float(s[0]), float(c[0])


def primary_ray(camera, W, H, x, y):
    cam = np.ascontiguousarray(camera, dtype=np.float32)
    o = np.zeros(3, np.float32)
    d = np.zeros(3, np.float32)
    lib.or_primary_ray(_p(cam), W, H, x, y, _p(o), _p(d))
    return o, d


def invert_transform(m):
    m = np.ascontiguousarray(m, dtype=np.float32).reshape(12)
    w = np.zeros(12, np.float32)
    lib.or_invert_transform(_p(m), _p(w))
    return w


class Blas:
    def __init__(self, vertices, indices, prim_count, max_leaf=4, max_vertex=None):
        """vertices: abi.VERTEX array (kept alive); indices: uint32 view starting at the mesh's first index."""
        self.vertices, self.indices = vertices, indices
        self.prim_count = prim_count
        self.h = lib.or_blas_build(_p(vertices), abi.VERTEX.itemsize, (len(vertices) - 1) if max_vertex is None else max_vertex,
                                   _p(indices), prim_count, max_leaf)

    def nodes(self):
        n = lib.or_blas_node_count(self.h)
        return np.ctypeslib.as_array(C.cast(lib.or_blas_nodes(self.h), C.POINTER(C.c_uint8)), shape=(n * 64,)).view(abi.BVH_NODE).copy()

    def leaf_prims(self):
        return np.ctypeslib.as_array(C.cast(lib.or_blas_leaf_prims(self.h), C.POINTER(C.c_uint32)), shape=(self.prim_count,)).copy()

    def sorted_keys(self):
        return np.ctypeslib.as_array(C.cast(lib.or_blas_sorted_keys(self.h), C.POINTER(C.c_uint64)), shape=(self.prim_count,)).copy()

    def root_box(self):
        lo = np.zeros(3, np.float32)
        hi = np.zeros(3, np.float32)
        lib.or_blas_root_box(self.h, _p(lo), _p(hi))
        return lo, hi

    def delta(self):
        return float(lib.or_blas_delta(self.h))

    def __del__(self):
        if getattr(self, "h", None):
            lib.or_blas_destroy(self.h)
            self.h = None


class Tlas:
    def __init__(self, instance_records, blases):
        self.records = np.ascontiguousarray(instance_records, dtype=abi.INSTANCE)
        self.blases = blases
        self.n = len(self.records)
        self.h = lib.or_tlas_build(_p(self.records), self.n)

    def nodes(self):
        n = lib.or_tlas_node_count(self.h)
        return np.ctypeslib.as_array(C.cast(lib.or_tlas_nodes(self.h), C.POINTER(C.c_uint8)), shape=(n * 64,)).view(abi.BVH_NODE).copy()

    def leaf_instances(self):
        return np.ctypeslib.as_array(C.cast(lib.or_tlas_leaf_instances(self.h), C.POINTER(C.c_uint32)), shape=(self.n,)).copy()

    def instance_box(self, i):
        lo = np.zeros(3, np.float32)
        hi = np.zeros(3, np.float32)
        lib.or_tlas_instance_box(self.h, i, _p(lo), _p(hi))
        return lo, hi

    def instance_w2o(self, i):
        w = np.zeros(12, np.float32)
        lib.or_tlas_instance_w2o(self.h, i, _p(w))
        return w

    def intersect(self, origins, dirs, tmin=0.001, tmax=10000.0, use_bvh=True, want_stats=False):
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        n = o.shape[0]
        hits = np.zeros(n, dtype=abi.HIT)
        t = np.zeros(n, dtype=np.float32)
        st = np.zeros(1, dtype=STATS)
        lib.or_intersect(self.h, n, _p(o), _p(d), tmin, tmax, 1 if use_bvh else 0, _p(hits), _p(t), _p(st))
        return (hits, t, st[0]) if want_stats else (hits, t)

    def __del__(self):
        if getattr(self, "h", None):
            lib.or_tlas_destroy(self.h)
            self.h = None


class OracleScene:
    """Builds BLAS/TLAS and the bindless tables on the CPU in Renderer::InitializeBLAS order
    (source/renderer.cpp:457-470, source/top_level_acceleration_structure.cpp:19-43)."""

    def __init__(self, scene, max_leaf=4):
        self.scene = scene
        self.blases, recs, geo = [], [], []
        self._keep = []
        for i, inst in enumerate(scene.instances()):
            model = scene.models[inst.model]
            mesh = model.meshes[inst.mesh]
            idx_view = model.indices[mesh.first_index:]
            blas = Blas(model.vertices, idx_view, mesh.index_count // 3, max_leaf)
            self.blases.append(blas)
            recs.append(abi.make_instance(inst.world[:3, :], i, blas.h))
            g = np.zeros((), dtype=abi.GEOMETRY_NODE)
            g["vertexBufferDeviceAddress"] = model.vertices.ctypes.data
            g["indexBufferDeviceAddress"] = model.indices.ctypes.data + 4 * mesh.first_index
            g["materialIndex"] = mesh.material
            geo.append(g)
        self.records = np.stack(recs).astype(abi.INSTANCE)
        self.tlas = Tlas(self.records, self.blases)
        self.materials = scene.tables.material_array()
        self.geometry_nodes = np.stack(geo).astype(abi.GEOMETRY_NODE)
        self.blas_instances = np.arange(len(geo), dtype=np.uint32).view(abi.BLAS_INSTANCE)
        self.tex = np.zeros(len(scene.tables.images), dtype=abi.TEXTURE_DESC)
        for k, img in enumerate(scene.tables.images):
            self.tex[k] = (img.ctypes.data, img.shape[1], img.shape[0])
        self.h = lib.or_scene_create(self.tlas.h, _p(self.materials), len(self.materials),
                                     _p(self.geometry_nodes), len(self.geometry_nodes),
                                     _p(self.blas_instances), len(self.blas_instances),
                                     _p(self.tex), len(self.tex))

    def render_frame(self, camera, W, H, frame=0, spp=16, max_depth=10, rect=None, use_bvh=True, threads=1):
        cam = np.ascontiguousarray(camera, dtype=np.float32)
        x0, y0, x1, y1 = rect if rect else (0, 0, W, H)
        res = np.zeros((H, W, 3), dtype=np.float32)
        st = np.zeros(1, dtype=STATS)
        lib.or_render_frame(self.h, _p(cam), W, H, frame, spp, max_depth, x0, y0, x1, y1, 1 if use_bvh else 0, threads, _p(res), _p(st))
        return res, st[0]

    def trace_path(self, camera, W, H, x, y, sample_index, max_depth=10):
        cam = np.ascontiguousarray(camera, dtype=np.float32)
        b = np.zeros(max_depth, dtype=BOUNCE)
        rad = np.zeros(3, np.float32)
        n = lib.or_trace_path(self.h, _p(cam), W, H, x, y, sample_index, max_depth, _p(b), _p(rad))
        return b[:n], rad

    def __del__(self):
        if getattr(self, "h", None):
            lib.or_scene_destroy(self.h)
            self.h = None


def accumulate(mode, frame, result, accum):
    r = np.ascontiguousarray(result, dtype=np.float32).reshape(-1, 3)
    lib.or_accumulate(mode, r.shape[0], frame, _p(r), _p(accum))


def resolve(accum_sum):
    a = np.ascontiguousarray(accum_sum, dtype=np.float32).reshape(-1, 4)
    out = np.zeros((a.shape[0], 4), dtype=np.uint8)
    lib.or_resolve(a.shape[0], _p(a), _p(out))
    return out


def texture_bilinear(img, u, v):
    d = np.zeros(1, dtype=abi.TEXTURE_DESC)
    d[0] = (img.ctypes.data, img.shape[1], img.shape[0])
    return [float(lib.or_texture_bilinear_channel(_p(d), np.float32(u), np.float32(v), k)) for k in range(4)]
