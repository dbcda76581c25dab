"""The shared-primary option (pt_ctx_set_share_primary): the reference traces the same primary ray for every
sample of a pixel (shaders/ray_gen.rgen:21-27,38-47), so bounce 0 may be traversed once per pixel.  Images must be
bit-identical to the default path and to the CPU oracle; the ray count drops by (spp - 1) x pixels."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _float_eq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


@pytest.mark.parametrize("schedule", ["plain", "auto"])
@pytest.mark.parametrize("name,w,h,spp", [("cornell", 128, 128, 16), ("cube", 96, 64, 8), ("dragon", 160, 90, 4), ("helmet", 96, 64, 3),
                                          ("cornell", 37, 23, 25), ("dragon", 53, 31, 1), ("cornell", 19, 11, 77)])
def test_shared_primary_frame_bit_exact(name, w, h, spp, schedule, request, monkeypatch):
    monkeypatch.setenv("PT_HIP_TRAVERSAL", schedule)
    import oracle as O
    from pt_amd import camera, hip
    scene = request.getfixturevalue(name)
    r = hip.Renderer(scene, w, h)
    try:
        r.ctx.set_share_primary(True)
        cam = camera.camera_uniform(w, h)
        orc = O.OracleScene(scene)
        rays_want = 0
        for frame in (0, 1):
            r.render(spp=spp, max_depth=10, frame=frame)
            got = r.last_result()
            want, st = orc.render_frame(cam, w, h, frame=frame, spp=spp, threads=8)
            assert _float_eq(got, want), f"frame {frame}: {np.sum(got != want)} of {got.size} values differ"
            rays_want += int(st["rays"]) - w * h * (spp - 1)
        s = r.stats()
        assert int(s["paths"]) == 2 * w * h * spp
        assert int(s["rays_per_depth"][0]) == 2 * w * h, "bounce 0 counts the rays that were traversed: one per pixel"
        assert int(s["rays"]) == rays_want
    finally:
        r.close()


@pytest.mark.parametrize("mode", [0, 2])
def test_shared_primary_chunks_and_frames_in_flight(mode, cornell):
    """Pixel chunks (the pixel hits live at the start of each chunk's hit records) and several frames in flight,
    with the option switched between frames: the same accumulated image as the default path."""
    from pt_amd import hip
    w, h = 200, 120
    images = {}
    for shared in (False, True):
        r = hip.Renderer(cornell, w, h)
        try:
            r.ctx.set_frames_in_flight(3)
            r.ctx.set_max_paths(30_000)          # four chunks per frame, the last partial
            for frame in range(5):
                r.ctx.set_share_primary(shared and frame != 2)      # frame 2 of the shared run uses the default path
                r.render(spp=5, frame=frame, accum_mode=mode)
            images[shared] = r.accum_host().copy()
        finally:
            r.close()
    assert np.array_equal(images[False].view(np.uint8), images[True].view(np.uint8))


def test_shared_primary_full_size_strip(dragon):
    """1920x1080, 64 spp in one chunk: a strip of the frame against the oracle and the whole frame against the default path."""
    import oracle as O
    from pt_amd import camera, hip
    W, H, spp = 1920, 1080, 64
    r = hip.Renderer(dragon, W, H)
    try:
        r.ctx.set_frames_in_flight(1)
        r.ctx.set_max_paths(W * H * spp)
        r.render(spp=spp, frame=7)
        base = r.last_result().copy()
        rays_default = int(r.stats()["rays"])
        r.reset_stats()
        r.ctx.set_share_primary(True)
        r.render(spp=spp, frame=7)
        got = r.last_result()
        assert _float_eq(got, base)
        assert int(r.stats()["rays"]) == rays_default - W * H * (spp - 1)
        x0, y0, x1, y1 = strip = (800, 500, 928, 508)
        want, st = O.OracleScene(dragon).render_frame(camera.camera_uniform(W, H), W, H, frame=7, spp=spp, rect=strip, threads=16)
        assert int(st["hits"]) > (x1 - x0) * (y1 - y0)
        assert _float_eq(got[y0:y1, x0:x1], want[y0:y1, x0:x1])
    finally:
        r.close()
