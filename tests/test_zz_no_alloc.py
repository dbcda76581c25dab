"""pt_intersect_noalloc (the entry point for callers that cannot allocate device memory) against
pt_intersect, through the C ABI: hit ids, barycentrics and distances bit for bit, on two calls in a
row of which the second is smaller (the fixed workspace keeps the first call's contents), plus the
size limit.  Collected last: a failure here must not hide the other GPU tests."""
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
