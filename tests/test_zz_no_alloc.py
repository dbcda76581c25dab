"""pt_intersect_noalloc (the entry point for callers that cannot allocate device memory) through the C ABI:
against pt_intersect (hit ids, barycentrics and distances bit for bit, on two calls in a row of which the second
is smaller: the fixed workspace keeps the first call's contents), against the CPU oracle directly, the size limit,
free device memory unchanged across calls, and one call recorded under stream capture and replayed as a graph.
Collected last: a failure here must not hide the other GPU tests."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rays(n, seed):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-2.5, 3.0, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d.astype(np.float32)


def _noalloc(r, o, d, tmin=0.001, tmax=10000.0):
    """The caller still owns inputs and outputs: they are allocated here, outside the call under test."""
    from pt_amd import abi, hip
    n = o.shape[0]
    ctx = r.ctx
    od, dd = ctx.upload(o), ctx.upload(d)
    hd, td = ctx.upload(np.full(16 * n, 0xAB, dtype=np.uint8)), ctx.upload(np.full(4 * n, 0xCD, dtype=np.uint8))
    hip._check(ctx.lib.pt_intersect_noalloc(ctx.h, r.tlas.h, n, od, dd, tmin, tmax, hd, td))
    hits = ctx.download(hd, 16 * n).view(abi.HIT).copy()
    t = ctx.download(td, 4 * n).view(np.float32).copy()
    for b in (od, dd, hd, td):
        ctx.free(b)
    return hits, t


@pytest.mark.parametrize("schedule", ["plain", "sched"])
def test_noalloc_matches_pt_intersect_on_two_calls(schedule, dragon, monkeypatch):
    monkeypatch.setenv("PT_HIP_TRAVERSAL", schedule)
    from pt_amd import hip
    r = hip.Renderer(dragon, 8, 8)
    try:
        for n, seed in ((300000, 5), (70001, 6)):      # the second call is smaller and its last batch is partial
            o, d = _rays(n, seed)
            want_h, want_t = r.intersect(o, d)
            got_h, got_t = _noalloc(r, o, d)
            assert np.array_equal(got_h, want_h), f"{n} rays: hit records differ"
            assert np.array_equal(got_t.view(np.uint32), want_t.view(np.uint32)), f"{n} rays: distances differ"
            assert (got_h["instance"] != 0xFFFFFFFF).mean() > 0.02
    finally:
        r.close()


def test_noalloc_rejects_more_rays_than_its_workspace_holds(cornell):
    from pt_amd import hip
    r = hip.Renderer(cornell, 8, 8)
    try:
        lib, ctx = r.ctx.lib, r.ctx
        buf = ctx.upload(np.zeros(64, dtype=np.uint8))
        rc = lib.pt_intersect_noalloc(ctx.h, r.tlas.h, (1 << 20) + 1, buf, buf, 0.001, 1e4, buf, None)
        assert rc == 4 and b"exceed" in lib.pt_last_error()      # PT_ERROR_LIMIT, nothing launched
        o, d = _rays(1 << 20, 9)                                   # the limit itself is accepted
        got_h, got_t = _noalloc(r, o, d)
        want_h, want_t = r.intersect(o, d)
        assert np.array_equal(got_h, want_h) and np.array_equal(got_t.view(np.uint32), want_t.view(np.uint32))
    finally:
        r.close()


@pytest.mark.parametrize("name,n", [("cornell", 150000), ("soup", 60000)])
def test_noalloc_matches_the_oracle(name, n, request):
    """Against the CPU oracle itself (its BVH traversal, and brute force over all triangles on a subset),
    not only against the entry point it shares kernels with."""
    import oracle as O
    from pt_amd import hip
    scene = request.getfixturevalue(name)
    r = hip.Renderer(scene, 8, 8)
    try:
        o, d = _rays(n, 21)
        got_h, got_t = _noalloc(r, o, d)
        orc = O.OracleScene(scene)
        want_h, want_t = orc.tlas.intersect(o, d, use_bvh=True)
        assert np.array_equal(got_h, want_h) and np.array_equal(got_t.view(np.uint32), want_t.view(np.uint32))
        m = 5000
        brute_h, brute_t = orc.tlas.intersect(o[:m], d[:m], use_bvh=False)
        assert np.array_equal(got_h[:m], brute_h) and np.array_equal(got_t[:m].view(np.uint32), brute_t.view(np.uint32))
    finally:
        r.close()


def _hip_runtime():
    rt = C.CDLL("libamdhip64.so")
    rt.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    rt.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    rt.hipStreamDestroy.argtypes = [C.c_void_p]
    rt.hipStreamSynchronize.argtypes = [C.c_void_p]
    rt.hipStreamBeginCapture.argtypes = [C.c_void_p, C.c_int]
    rt.hipStreamEndCapture.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    rt.hipGraphInstantiate.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    rt.hipGraphLaunch.argtypes = [C.c_void_p, C.c_void_p]
    rt.hipGraphExecDestroy.argtypes = [C.c_void_p]
    rt.hipGraphDestroy.argtypes = [C.c_void_p]
    rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    return rt


def test_noalloc_leaves_the_free_memory_as_it_was(cornell):
    """hipMemGetInfo before and after a series of calls of different sizes: not a byte is taken or given back."""
    from pt_amd import hip
    rt = _hip_runtime()
    r = hip.Renderer(cornell, 8, 8)
    try:
        ctx = r.ctx
        n_max = 400000
        o, d = _rays(n_max, 31)
        od, dd = ctx.upload(o), ctx.upload(d)
        hd, td = ctx.upload(np.zeros(16 * n_max, dtype=np.uint8)), ctx.upload(np.zeros(4 * n_max, dtype=np.uint8))
        hip._check(ctx.lib.pt_intersect_noalloc(ctx.h, r.tlas.h, 1000, od, dd, 0.001, 1e4, hd, td))      # the first launch loads the kernels
        ctx.sync()
        free0, free1, total = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        assert rt.hipMemGetInfo(C.byref(free0), C.byref(total)) == 0
        for n in (n_max, 17, 250000, 64, 99999):
            hip._check(ctx.lib.pt_intersect_noalloc(ctx.h, r.tlas.h, n, od, dd, 0.001, 1e4, hd, td))
            assert rt.hipMemGetInfo(C.byref(free1), C.byref(total)) == 0       # calls still in flight
            assert free1.value == free0.value
        ctx.sync()
        assert rt.hipMemGetInfo(C.byref(free1), C.byref(total)) == 0
        assert free1.value == free0.value
    finally:
        r.close()


def test_noalloc_under_stream_capture_and_graph_replay(cornell):
    """The call is recorded into a HIP graph under global-mode capture (where any allocation or synchronising call
    of the process would end the capture with an error) and the graph is replayed on new rays: both replays agree
    with the oracle."""
    import oracle as O
    from pt_amd import abi, hip
    rt = _hip_runtime()
    stream = C.c_void_p()
    assert rt.hipStreamCreate(C.byref(stream)) == 0
    r = hip.Renderer(cornell, 8, 8, stream=stream.value)
    graph, gexec = C.c_void_p(), C.c_void_p()
    try:
        ctx = r.ctx
        n = 200000
        o, d = _rays(n, 41)
        od, dd = ctx.upload(o), ctx.upload(d)
        hd, td = ctx.upload(np.zeros(16 * n, dtype=np.uint8)), ctx.upload(np.zeros(4 * n, dtype=np.uint8))
        hip._check(ctx.lib.pt_intersect_noalloc(ctx.h, r.tlas.h, 64, od, dd, 0.001, 1e4, hd, td))        # kernels loaded before the capture
        ctx.sync()
        assert rt.hipStreamBeginCapture(stream, 0) == 0                # hipStreamCaptureModeGlobal
        rc = ctx.lib.pt_intersect_noalloc(ctx.h, r.tlas.h, n, od, dd, 0.001, 1e4, hd, td)
        end = rt.hipStreamEndCapture(stream, C.byref(graph))
        assert rc == 0, ctx.lib.pt_last_error()
        assert end == 0 and graph.value, "the capture was invalidated: the call did something a capture forbids"
        assert rt.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0) == 0
        orc = O.OracleScene(cornell)
        for seed in (41, 42):
            o, d = _rays(n, seed)
            assert rt.hipMemcpy(od, o.ctypes.data, o.nbytes, 1) == 0 and rt.hipMemcpy(dd, d.ctypes.data, d.nbytes, 1) == 0
            assert rt.hipGraphLaunch(gexec, stream) == 0
            assert rt.hipStreamSynchronize(stream) == 0
            got_h = ctx.download(hd, 16 * n).view(abi.HIT).copy()
            got_t = ctx.download(td, 4 * n).view(np.float32).copy()
            want_h, want_t = orc.tlas.intersect(o, d, use_bvh=True)
            assert np.array_equal(got_h, want_h) and np.array_equal(got_t.view(np.uint32), want_t.view(np.uint32))
    finally:
        if gexec.value:
            rt.hipGraphExecDestroy(gexec)
        if graph.value:
            rt.hipGraphDestroy(graph)
        r.close()
        rt.hipStreamDestroy(stream)
