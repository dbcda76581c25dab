"""ctypes binding of libpt_hip.so (the C ABI of include/pt_hip.h) and a Renderer that drives it.

`Renderer` mirrors the reference's Renderer (include/renderer.hpp:20-89): its constructor loads
the models, builds one BLAS per (node, mesh) and the TLAS (source/renderer.cpp:13-42,457-470) and
uploads the bindless tables; `render()` is Renderer::Render (source/renderer.cpp:60-107) without
the swapchain: one pt_trace per call, `_rendered_frames` incremented after each.
The HIP library is required: there is no CPU fallback in this package.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PT_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libpt_hip.so")

PT_MAX_DEPTH = 64
SEED_SAMPLE_INDEX, SEED_CLOCK = 0, 1
ACCUM_SUM_F32, ACCUM_MEAN_F32, ACCUM_MEAN_UNORM8 = 0, 1, 2

AS_INFO = np.dtype([("node_count", "<u4"), ("prim_count", "<u4"), ("root_lo", "<f4", 3), ("root_hi", "<f4", 3),
                    ("delta", "<f4"), ("build_ms", "<f4")])
TRACE_PROFILE = np.dtype([("intersect_launches", "<u8"), ("intersect_rays", "<u8"), ("intersect_ms", "<f8"),
                          ("primary_launches", "<u8"), ("primary_rays", "<u8"), ("primary_ms", "<f8")])
TRACE_STATS = np.dtype([("rays_per_depth", "<u8", PT_MAX_DEPTH), ("rays", "<u8"), ("paths", "<u8")])

# every symbol include/pt_hip.h declares: (restype, argtypes)
_vp, _u32, _u64, _f, _i, _sz = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float, C.c_int, C.c_size_t
SIGNATURES = {
    "pt_last_error": (C.c_char_p, []),
    "pt_ctx_create": (_i, [_i, _vp, C.POINTER(_vp)]),
    "pt_ctx_destroy": (_i, [_vp]),
    "pt_sync": (_i, [_vp]),
    "pt_ctx_set_frames_in_flight": (_i, [_vp, C.c_int]),
    "pt_trace_fence": (_i, [_vp]),
    "pt_buffer_upload": (_i, [_vp, _vp, _sz, C.POINTER(_vp)]),
    "pt_buffer_download": (_i, [_vp, _vp, _sz, _vp]),
    "pt_buffer_free": (_i, [_vp, _vp]),
    "pt_blas_build": (_i, [_vp, _vp, _u32, _u32, _vp, _u32, _u32, C.POINTER(_vp)]),
    "pt_blas_destroy": (_i, [_vp]),
    "pt_blas_info": (_i, [_vp, _vp]),
    "pt_blas_download": (_i, [_vp, _vp, _vp, _vp]),
    "pt_blas_wide_download": (_i, [_vp, _vp, _u32, _vp, _vp]),
    "pt_blas_traversal_download": (_i, [_vp, _vp, _u32, _vp, _vp]),
    "pt_tlas_wide_download": (_i, [_vp, _vp, _u32, _vp, _vp]),
    "pt_tlas_traversal_download": (_i, [_vp, _vp, _u32, _vp, _vp]),
    "pt_tlas_build": (_i, [_vp, _vp, _u32, C.POINTER(_vp)]),
    "pt_tlas_destroy": (_i, [_vp]),
    "pt_tlas_info": (_i, [_vp, _vp]),
    "pt_tlas_download": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "pt_scene_set_tables": (_i, [_vp, _vp, _u32, _vp, _u32, _vp, _u32, _vp, _u32]),
    "pt_intersect": (_i, [_vp, _vp, _u32, _vp, _vp, _f, _f, _vp, _vp]),
    "pt_intersect_noalloc": (_i, [_vp, _vp, _u32, _vp, _vp, _f, _f, _vp, _vp]),
    "pt_trace": (_i, [_vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _vp]),
    "pt_trace_last_result": (_i, [_vp, _vp]),
    "pt_trace_stats": (_i, [_vp, _vp]),
    "pt_trace_stats_reset": (_i, [_vp]),
    "pt_ctx_set_max_paths": (_i, [_vp, _u64]),
    "pt_ctx_set_profiling": (_i, [_vp, _i]),
    "pt_ctx_set_share_primary": (_i, [_vp, _i]),
    "pt_trace_profile": (_i, [_vp, _vp]),
    "pt_accum_resolve": (_i, [_vp, _vp, _u32, _vp, _i]),
    "pt_accum_allreduce": (_i, [_vp, _vp, _u32, _u32]),
    "pt_selftest_arith": (_i, [_vp, _vp]),
}

_lib = None


class PtError(RuntimeError):
    pass


def load_library():
    """Loads libpt_hip.so; raises if it has not been built (there is no fallback path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PtError(f"{LIB_PATH} is missing: build it with `make -C path-tracer_amd/csrc`")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def _check(rc):
    if rc != 0:
        raise PtError(f"pt status {rc}: {load_library().pt_last_error().decode()}")


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    def __init__(self, device=0, stream=None):
        self.lib = load_library()
        h = C.c_void_p()
        _check(self.lib.pt_ctx_create(device, stream, C.byref(h)))
        self.h = h
        self._buffers = []

    def upload(self, array):
        a = np.ascontiguousarray(array)
        d = C.c_void_p()
        _check(self.lib.pt_buffer_upload(self.h, _p(a), a.nbytes, C.byref(d)))
        self._buffers.append(d.value)
        return d.value

    def alloc_zeros(self, nbytes):
        return self.upload(np.zeros(nbytes, dtype=np.uint8))

    def download(self, dev, nbytes, dtype=np.uint8):
        out = np.empty(nbytes, dtype=np.uint8)
        _check(self.lib.pt_buffer_download(self.h, dev, nbytes, _p(out)))
        return out.view(dtype)

    def free(self, dev):
        if dev in self._buffers:
            self._buffers.remove(dev)
            _check(self.lib.pt_buffer_free(self.h, dev))

    def sync(self):
        _check(self.lib.pt_sync(self.h))

    def set_frames_in_flight(self, n):
        _check(self.lib.pt_ctx_set_frames_in_flight(self.h, n))

    def fence(self):
        """The context's stream waits for every frame in flight (no host wait)."""
        _check(self.lib.pt_trace_fence(self.h))

    def set_max_paths(self, n):
        _check(self.lib.pt_ctx_set_max_paths(self.h, n))

    def set_share_primary(self, enabled):
        """Bounce 0 traced once per pixel and shared by the pixel's samples (identical images, fewer rays traversed)."""
        _check(self.lib.pt_ctx_set_share_primary(self.h, int(enabled)))

    def selftest_arith(self):
        """pt_selftest_arith: results of the short forms of 1/x, sqrt, x/pi, byte/255 and the index division that
        differ from the IEEE operations over all inputs, as a dict of counts (all must be 0)."""
        out = (C.c_uint64 * 5)()
        _check(self.lib.pt_selftest_arith(self.h, out))
        return dict(zip(("reciprocal", "sqrt", "div_pi", "unorm8", "index_division"), (int(v) for v in out)))

    def close(self):
        if self.h:
            for d in list(self._buffers):
                self.lib.pt_buffer_free(self.h, d)
            self._buffers = []
            self.lib.pt_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def accum_allreduce(contexts, accum_ptrs, n_pixels):
    """pt_accum_allreduce over several contexts of this process (one per GPU): every float sum image ends up holding the total."""
    n = len(contexts)
    cs = (C.c_void_p * n)(*[c.h for c in contexts])
    ps = (C.c_void_p * n)(*accum_ptrs)
    _check(load_library().pt_accum_allreduce(cs, ps, n, n_pixels))


class BottomLevelAccelerationStructure:
    """include/bottom_level_acceleration_structure.hpp:20-37."""

    def __init__(self, ctx, vertices_dev, vertex_stride, max_vertex, indices_dev, prim_count, transform, max_leaf=0):
        self.ctx = ctx
        self._transform = np.asarray(transform, dtype=np.float32)
        h = C.c_void_p()
        _check(ctx.lib.pt_blas_build(ctx.h, vertices_dev, vertex_stride, max_vertex, indices_dev, prim_count, max_leaf, C.byref(h)))
        self.h = h
        self.prim_count = prim_count

    def structure(self):
        return self.h.value

    def transform(self):
        return self._transform

    def info(self):
        out = np.zeros(1, dtype=AS_INFO)
        _check(self.ctx.lib.pt_blas_info(self.h, _p(out)))
        return out[0]

    def download(self):
        """The Morton-order tree (built by the first call: the kernels traverse the SAH tree)."""
        _check(self.ctx.lib.pt_blas_download(self.h, None, None, None))
        info = self.info()
        nodes = np.zeros(int(info["node_count"]), dtype=abi.BVH_NODE)
        prims = np.zeros(self.prim_count, dtype=np.uint32)
        keys = np.zeros(self.prim_count, dtype=np.uint64)
        _check(self.ctx.lib.pt_blas_download(self.h, _p(nodes), _p(prims), _p(keys)))
        return nodes, prims, keys

    def download_traversal_tree(self):
        """(binary nodes, primitive per leaf position) of the tree the wide nodes are derived from."""
        cnt = C.c_uint32(0)
        _check(self.ctx.lib.pt_blas_traversal_download(self.h, None, 0, None, C.byref(cnt)))
        nodes = np.zeros(cnt.value, dtype=abi.BVH_NODE)
        prims = np.zeros(self.prim_count, dtype=np.uint32)
        _check(self.ctx.lib.pt_blas_traversal_download(self.h, _p(nodes), cnt.value, _p(prims), None))
        return nodes, prims

    def download_wide(self):
        """(wide nodes, depth) of the traversal layout."""
        cnt, depth = C.c_uint32(0), C.c_uint32(0)
        _check(self.ctx.lib.pt_blas_wide_download(self.h, None, 0, C.byref(cnt), C.byref(depth)))
        nodes = np.zeros(cnt.value, dtype=abi.WIDE_NODE)
        _check(self.ctx.lib.pt_blas_wide_download(self.h, _p(nodes), cnt.value, None, None))
        return nodes, depth.value

    def destroy(self):
        if self.h:
            self.ctx.lib.pt_blas_destroy(self.h)
            self.h = None


class TopLevelAccelerationStructure:
    """include/top_level_acceleration_structure.hpp:9-23."""

    def __init__(self, ctx, blases):
        self.ctx = ctx
        self.n = len(blases)
        recs = np.stack([abi.make_instance(b.transform()[:3, :], i, b.structure()) for i, b in enumerate(blases)]).astype(abi.INSTANCE)
        self.records = recs
        h = C.c_void_p()
        _check(ctx.lib.pt_tlas_build(ctx.h, _p(recs), self.n, C.byref(h)))
        self.h = h

    def structure(self):
        return self.h.value

    def info(self):
        out = np.zeros(1, dtype=AS_INFO)
        _check(self.ctx.lib.pt_tlas_info(self.h, _p(out)))
        return out[0]

    def download(self):
        info = self.info()
        nodes = np.zeros(int(info["node_count"]), dtype=abi.BVH_NODE)
        leaf = np.zeros(self.n, dtype=np.uint32)
        boxes = np.zeros((self.n, 6), dtype=np.float32)
        w2o = np.zeros((self.n, 12), dtype=np.float32)
        _check(self.ctx.lib.pt_tlas_download(self.h, _p(nodes), _p(leaf), _p(boxes), _p(w2o)))
        return nodes, leaf, boxes, w2o

    def download_traversal_tree(self):
        """(binary nodes, instance per leaf position) of the tree the wide nodes are derived from."""
        cnt = C.c_uint32(0)
        _check(self.ctx.lib.pt_tlas_traversal_download(self.h, None, 0, None, C.byref(cnt)))
        nodes = np.zeros(cnt.value, dtype=abi.BVH_NODE)
        inst = np.zeros(self.n, dtype=np.uint32)
        _check(self.ctx.lib.pt_tlas_traversal_download(self.h, _p(nodes), cnt.value, _p(inst), None))
        return nodes, inst

    def download_wide(self):
        """(wide nodes, depth) of the traversal layout."""
        cnt, depth = C.c_uint32(0), C.c_uint32(0)
        _check(self.ctx.lib.pt_tlas_wide_download(self.h, None, 0, C.byref(cnt), C.byref(depth)))
        nodes = np.zeros(cnt.value, dtype=abi.WIDE_NODE)
        _check(self.ctx.lib.pt_tlas_wide_download(self.h, _p(nodes), cnt.value, None, None))
        return nodes, depth.value

    def destroy(self):
        if self.h:
            self.ctx.lib.pt_tlas_destroy(self.h)
            self.h = None


class Renderer:
    """Headless equivalent of the reference Renderer (include/renderer.hpp:20-89)."""

    SAMPLES_PER_FRAME = 25   # shaders/ray_gen.rgen:34
    MAX_DEPTH = 10           # shaders/ray_gen.rgen:52

    def __init__(self, scene, width, height, device=0, stream=None, max_leaf=0, camera=None):
        from . import camera as cam_mod
        self.ctx = Context(device, stream)
        self.scene = scene
        self.width, self.height = width, height
        self._rendered_frames = 0
        ctx = self.ctx
        # model buffers (ModelLoader::ProcessModel uploads, source/model_loader.cpp:307-357)
        self._model_bufs = [(ctx.upload(m.vertices), ctx.upload(m.indices)) for m in scene.models]
        # InitializeBLAS, source/renderer.cpp:457-470
        self.blases, geo = [], []
        for inst in scene.instances():
            model = scene.models[inst.model]
            mesh = model.meshes[inst.mesh]
            vdev, idev = self._model_bufs[inst.model]
            iaddr = idev + 4 * mesh.first_index
            self.blases.append(BottomLevelAccelerationStructure(ctx, vdev, abi.VERTEX.itemsize, len(model.vertices) - 1, iaddr,
                                                                mesh.index_count // 3, inst.world, max_leaf))
            g = np.zeros((), dtype=abi.GEOMETRY_NODE)
            g["vertexBufferDeviceAddress"], g["indexBufferDeviceAddress"], g["materialIndex"] = vdev, iaddr, mesh.material
            geo.append(g)
        self.tlas = TopLevelAccelerationStructure(ctx, self.blases)
        # BindlessResources::UpdateDescriptorSet
        self.materials = scene.tables.material_array()
        self.geometry_nodes = np.stack(geo).astype(abi.GEOMETRY_NODE)
        self.blas_instances = np.arange(len(geo), dtype=np.uint32).view(abi.BLAS_INSTANCE)
        self.textures = np.zeros(len(scene.tables.images), dtype=abi.TEXTURE_DESC)
        for k, img in enumerate(scene.tables.images):
            self.textures[k] = (ctx.upload(img), img.shape[1], img.shape[0])
        _check(ctx.lib.pt_scene_set_tables(ctx.h, _p(self.materials), len(self.materials), _p(self.geometry_nodes), len(self.geometry_nodes),
                                           _p(self.blas_instances), len(self.blas_instances), _p(self.textures), len(self.textures)))
        self.camera = np.ascontiguousarray(camera if camera is not None else cam_mod.camera_uniform(width, height), dtype=np.float32)
        self._accum = None
        self._accum_mode = None

    # -- accumulation target (InitializeRenderTarget, source/renderer.cpp:161-170)
    def _ensure_accum(self, mode):
        if self._accum is None or self._accum_mode != mode:
            nbytes = self.width * self.height * (4 if mode == ACCUM_MEAN_UNORM8 else 16)
            self._accum = self.ctx.alloc_zeros(nbytes)
            self._accum_mode = mode
        return self._accum

    def render(self, spp=None, max_depth=None, accum_mode=ACCUM_SUM_F32, seed_mode=SEED_SAMPLE_INDEX, accum_dev=None, frame=None):
        """Renderer::Render: traces one frame and bumps the frame counter."""
        spp = spp or self.SAMPLES_PER_FRAME
        max_depth = max_depth or self.MAX_DEPTH
        accum = accum_dev if accum_dev is not None else self._ensure_accum(accum_mode)
        f = self._rendered_frames if frame is None else frame
        _check(self.ctx.lib.pt_trace(self.ctx.h, self.tlas.h, _p(self.camera), self.width, self.height, f, spp, max_depth,
                                     seed_mode, accum_mode, accum))
        self._rendered_frames += 1

    def last_result(self):
        out = np.zeros((self.height, self.width, 3), dtype=np.float32)
        _check(self.ctx.lib.pt_trace_last_result(self.ctx.h, _p(out)))
        return out

    def accum_host(self):
        if self._accum_mode == ACCUM_MEAN_UNORM8:
            return self.ctx.download(self._accum, self.width * self.height * 4, np.uint8).reshape(self.height, self.width, 4)
        return self.ctx.download(self._accum, self.width * self.height * 16, np.float32).reshape(self.height, self.width, 4)

    def resolve(self, bgra=False, accum_dev=None):
        out = self.ctx.alloc_zeros(self.width * self.height * 4)
        _check(self.ctx.lib.pt_accum_resolve(self.ctx.h, accum_dev if accum_dev is not None else self._accum, self.width * self.height, out, int(bgra)))
        img = self.ctx.download(out, self.width * self.height * 4).reshape(self.height, self.width, 4).copy()
        self.ctx.free(out)
        return img

    def stats(self):
        out = np.zeros(1, dtype=TRACE_STATS)
        _check(self.ctx.lib.pt_trace_stats(self.ctx.h, _p(out)))
        return out[0]

    def fence(self):
        """Orders everything traced so far before later work on the context's stream."""
        self.ctx.fence()

    def set_profiling(self, enabled):
        _check(self.ctx.lib.pt_ctx_set_profiling(self.ctx.h, int(enabled)))

    def profile(self):
        out = np.zeros(1, dtype=TRACE_PROFILE)
        _check(self.ctx.lib.pt_trace_profile(self.ctx.h, _p(out)))
        return out[0]

    def reset_stats(self):
        _check(self.ctx.lib.pt_trace_stats_reset(self.ctx.h))

    def intersect(self, origins, dirs, tmin=0.001, tmax=10000.0):
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, dtype=np.float32).reshape(-1, 3)
        n = o.shape[0]
        od, dd = self.ctx.upload(o), self.ctx.upload(d)
        hd, td = self.ctx.alloc_zeros(16 * n), self.ctx.alloc_zeros(4 * n)
        _check(self.ctx.lib.pt_intersect(self.ctx.h, self.tlas.h, n, od, dd, tmin, tmax, hd, td))
        hits = self.ctx.download(hd, 16 * n).view(abi.HIT).copy()
        t = self.ctx.download(td, 4 * n).view(np.float32).copy()
        for b in (od, dd, hd, td):
            self.ctx.free(b)
        return hits, t

    def close(self):
        if getattr(self, "tlas", None):
            self.tlas.destroy()
        for b in getattr(self, "blases", []):
            b.destroy()
        self.blases = []
        self.ctx.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
