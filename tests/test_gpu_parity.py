"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical inputs.
Integer results (BVH topology, hit ids) and, because sin/cos and every operation order are pinned,
all float results as well are compared bit for bit."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle(scene):
    import oracle as O
    return O.OracleScene(scene)


def _renderer(scene, w, h, **kw):
    from pt_amd import hip
    return hip.Renderer(scene, w, h, **kw)


def _random_rays(n, seed, lo=-2.5, hi=3.0):
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d.astype(np.float32)


def _float_eq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


@pytest.mark.parametrize("name", ["cornell", "cube", "soup", "dragon", "helmet"])
def test_bvh_topology_bit_exact(name, request):
    scene = request.getfixturevalue(name)
    orc = _oracle(scene)
    r = _renderer(scene, 8, 8)
    try:
        for i, (ob, gb) in enumerate(zip(orc.blases, r.blases)):
            if gb.prim_count > 1:       # the kernels traverse the SAH tree: the Morton tree exists from its first download on
                assert int(gb.info()["node_count"]) == 0
            nodes, prims, keys = gb.download()
            assert int(gb.info()["node_count"]) == len(nodes) > 0
            again = gb.download()
            assert all(np.array_equal(a.view(np.uint8), b.view(np.uint8)) for a, b in zip((nodes, prims, keys), again))
            assert np.array_equal(keys, ob.sorted_keys()), f"blas {i}: sorted Morton keys differ"
            assert np.array_equal(prims, ob.leaf_prims()), f"blas {i}: primitive order differs"
            on = ob.nodes()
            assert len(nodes) == len(on), f"blas {i}: node count {len(nodes)} vs {len(on)}"
            assert np.array_equal(nodes["c0"], on["c0"]) and np.array_equal(nodes["c1"], on["c1"]), f"blas {i}: child references differ"
            for f in ("lo0", "hi0", "lo1", "hi1"):
                assert _float_eq(nodes[f], on[f]), f"blas {i}: boxes differ in {f}"
            info = gb.info()
            lo, hi = ob.root_box()
            assert _float_eq(info["root_lo"], lo) and _float_eq(info["root_hi"], hi)
            assert float(info["delta"]) == ob.delta()
        nodes, leaf, boxes, w2o = r.tlas.download()
        on = orc.tlas.nodes()
        assert np.array_equal(leaf, orc.tlas.leaf_instances())
        assert len(nodes) == len(on)
        assert np.array_equal(nodes["c0"], on["c0"]) and np.array_equal(nodes["c1"], on["c1"])
        for f in ("lo0", "hi0", "lo1", "hi1"):
            assert _float_eq(nodes[f], on[f])
        for i in range(orc.tlas.n):
            lo, hi = orc.tlas.instance_box(i)
            assert _float_eq(boxes[i, :3], lo) and _float_eq(boxes[i, 3:], hi)
            assert _float_eq(w2o[i], orc.tlas.instance_w2o(i))
    finally:
        r.close()


@pytest.mark.parametrize("schedule", ["plain", "sched"])
@pytest.mark.parametrize("name,n", [("cornell", 200000), ("cube", 100000), ("soup", 100000), ("dragon", 100000)])
def test_intersect_hit_ids_bit_exact(name, n, schedule, request, monkeypatch):
    monkeypatch.setenv("PT_HIP_TRAVERSAL", schedule)   # both traversal schedules, whatever the level size
    scene = request.getfixturevalue(name)
    orc = _oracle(scene)
    r = _renderer(scene, 8, 8)
    try:
        o, d = _random_rays(n, seed=11)
        hits, t = r.intersect(o, d)
        oh, ot = orc.tlas.intersect(o, d, use_bvh=True)
        assert np.array_equal(hits["instance"], oh["instance"])
        assert np.array_equal(hits["primitive"], oh["primitive"])
        assert _float_eq(hits["u"], oh["u"]) and _float_eq(hits["v"], oh["v"]) and _float_eq(t, ot)
        assert (hits["instance"] != 0xFFFFFFFF).mean() > 0.02
        # brute force (no BVH at all) on a subset
        m = 3000 if name == "dragon" else 20000
        bh, bt = orc.tlas.intersect(o[:m], d[:m], use_bvh=False)
        assert np.array_equal(hits[:m], bh) and _float_eq(t[:m], bt)
    finally:
        r.close()


@pytest.mark.parametrize("schedule", ["plain", "sched"])
@pytest.mark.parametrize("name,w,h,spp", [("cornell", 128, 128, 16), ("cube", 96, 64, 8), ("dragon", 160, 90, 4), ("soup", 64, 64, 4),
                                          # the reference's 25 samples per frame on an odd image (batches straddle pixels, the last
                                          # batch is partial), one sample, and more samples than a batch holds
                                          ("cornell", 37, 23, 25), ("dragon", 53, 31, 1), ("cornell", 19, 11, 77)])
def test_render_frame_bit_exact(name, w, h, spp, schedule, request, monkeypatch):
    monkeypatch.setenv("PT_HIP_TRAVERSAL", schedule)
    from pt_amd import camera
    scene = request.getfixturevalue(name)
    orc = _oracle(scene)
    r = _renderer(scene, w, h)
    try:
        cam = camera.camera_uniform(w, h)
        for frame in (0, 1):
            r.render(spp=spp, max_depth=10, frame=frame)
            got = r.last_result()
            want, st = orc.render_frame(cam, w, h, frame=frame, spp=spp, threads=8)
            assert _float_eq(got, want), f"frame {frame}: {np.sum(got != want)} of {got.size} values differ"
        stats = r.stats()
        assert int(stats["paths"]) == 2 * w * h * spp
    finally:
        r.close()


@pytest.mark.parametrize("tw,th", [(37, 53), (64, 24), (1, 7)])
def test_texture_sizes_that_are_not_powers_of_two(tw, th, cube):
    """REPEAT addressing with the cube's uvs in [-1, 1] on textures whose sizes are not powers of two
    (the sampler wraps those by remainder, powers of two by mask)."""
    import copy
    from pt_amd import camera
    scene = copy.deepcopy(cube)
    tex = [i for i, im in enumerate(scene.tables.images) if i > 0 and im.shape[0] > 2]
    assert tex
    rng = np.random.default_rng(tw * 100 + th)
    for i in tex:
        scene.tables.images[i] = rng.integers(0, 256, size=(th, tw, 4), dtype=np.uint8)
    w, h, spp = 80, 60, 8
    orc = _oracle(scene)
    r = _renderer(scene, w, h)
    try:
        r.render(spp=spp, max_depth=10, frame=0)
        want, _ = orc.render_frame(camera.camera_uniform(w, h), w, h, frame=0, spp=spp, threads=8)
        assert _float_eq(r.last_result(), want)
    finally:
        r.close()


HELMET_CAMERA = dict(eye=(0.15, 0.1, 0.75), center=(0.0, 0.0, 0.0), fov_deg=50.0)


@pytest.mark.parametrize("schedule", ["plain", "sched"])
def test_helmet_textured_frame_bit_exact(helmet, schedule, monkeypatch):
    """BASELINE config 3's scene (FlightHelmet: 5 instances, 94,722 triangles, four base-colour
    textures sampled through the mesh's own UVs) with a camera that frames it: hit records of the
    primary rays and two frames, bit for bit; then 1080p for the sample-count bookkeeping."""
    monkeypatch.setenv("PT_HIP_TRAVERSAL", schedule)
    import oracle as O
    from pt_amd import camera
    w, h, spp = 240, 200, 8
    cam = camera.camera_uniform(w, h, **HELMET_CAMERA)
    orc = _oracle(helmet)
    r = _renderer(helmet, w, h, camera=cam)
    try:
        o = np.zeros((w * h, 3), np.float32); d = np.zeros((w * h, 3), np.float32)
        for y in range(h):
            for x in range(w):
                o[y * w + x], d[y * w + x] = O.primary_ray(cam, w, h, x, y)
        hg, tg = r.intersect(o, d)
        ho, to = orc.tlas.intersect(o, d)
        assert np.array_equal(hg, ho) and _float_eq(tg, to)
        hit = ho["instance"] != 0xFFFFFFFF
        assert hit.mean() > 0.1 and len(set(ho["instance"][hit].tolist())) >= 4
        for frame in (0, 1):
            r.render(spp=spp, max_depth=10, frame=frame)
            want, _ = orc.render_frame(cam, w, h, frame=frame, spp=spp, threads=8)
            assert _float_eq(r.last_result(), want)
        # textured surfaces are not grey: the base-colour maps really are sampled
        res = r.last_result()[hit.reshape(h, w)]
        assert np.mean(np.abs(res[:, 0] - res[:, 2])) > 1e-3
    finally:
        r.close()
    if schedule == "sched":
        W, H = 1920, 1080
        r = _renderer(helmet, W, H, camera=camera.camera_uniform(W, H, **HELMET_CAMERA))
        try:
            r.render(spp=4, frame=0)
            s = r.stats()
            assert int(s["paths"]) == W * H * 4 and int(s["rays_per_depth"][0]) == W * H * 4
            res = r.last_result()
            assert np.isfinite(res).all() and (res >= 0).all()
        finally:
            r.close()


def test_exact_ties_between_instances(stacked, cornell):
    """Ties in t are decided by (instance, primitive); every field of the winner must be reported.
    (a) identical stacked copies: the traversal meets the copies in tree order, the answer is copy 0;
    (b) a Cornell primary ray through the edge shared by the floor and the right wall, where both
    triangles give the same t but different barycentrics."""
    import oracle as O
    r = _renderer(stacked, 64, 64)
    try:
        orc = O.OracleScene(stacked)
        o, d = _random_rays(20000, 21, -2.0, 2.0)
        hg, tg = r.intersect(o, d)
        ho, to = orc.tlas.intersect(o, d)
        assert np.array_equal(hg, ho) and _float_eq(tg, to)
        hit = hg["instance"] != 0xFFFFFFFF
        assert hit.mean() > 0.1 and np.all(hg["instance"][hit] == 0)
    finally:
        r.close()
    r = _renderer(cornell, 64, 64)
    try:
        orc = O.OracleScene(cornell)
        o = np.array([[0, 1, 3]], np.float32)
        d = np.array([[1050287530, 3197771178, 3211240841]], np.uint32).view(np.float32)
        hg, tg = r.intersect(o, d)
        ho, to = orc.tlas.intersect(o, d)
        assert (int(ho["instance"][0]), int(ho["primitive"][0])) == (0, 0)
        assert np.array_equal(hg, ho) and _float_eq(tg, to)
    finally:
        r.close()


@pytest.mark.parametrize("schedule", ["plain", "sched", "auto"])
def test_cfg1_golden_full_size(cornell, schedule, monkeypatch):
    """BASELINE config 1 in full (Cornell 512x512, 16 spp, frame 0) against the committed oracle
    outputs: SHA-256 of the whole float image, the radiance crop, the ray count and the bounce-0 ids."""
    monkeypatch.setenv("PT_HIP_TRAVERSAL", schedule)
    import hashlib
    import os
    from conftest import GOLDEN
    from pt_amd import camera
    import oracle as O
    g = np.load(os.path.join(GOLDEN, "cfg1_cornell_512_16spp.npz"))
    W = H = 512
    r = _renderer(cornell, W, H)
    try:
        r.render(spp=16, max_depth=10, frame=0)
        got = r.last_result()
        assert int(r.stats()["rays"]) == int(g["rays"])
        assert np.array_equal(got[192:320, 192:320], g["crop"])
        assert hashlib.sha256(got.tobytes()).digest() == g["sha256"].tobytes()
        cam = camera.camera_uniform(W, H)
        o = np.zeros((H * W, 3), np.float32)
        d = np.zeros((H * W, 3), np.float32)
        for y in range(0, H, 4):               # every fourth row keeps the host loop short
            for x in range(W):
                o[y * W + x], d[y * W + x] = O.primary_ray(cam, W, H, x, y)
        rows = np.arange(0, H, 4)
        sel = (rows[:, None] * W + np.arange(W)[None, :]).reshape(-1)
        hits, _ = r.intersect(o[sel], d[sel])
        ids = g["bounce0_ids"].reshape(-1, 2)[sel]
        inst = np.where(hits["instance"] == 0xFFFFFFFF, 255, hits["instance"]).astype(np.uint8)
        prim = np.where(hits["primitive"] == 0xFFFFFFFF, 255, hits["primitive"]).astype(np.uint8)
        assert np.array_equal(inst, ids[:, 0]) and np.array_equal(prim, ids[:, 1])
    finally:
        r.close()


def test_ray_counts_match_oracle(cornell):
    from pt_amd import camera
    orc = _oracle(cornell)
    w = h = 64
    r = _renderer(cornell, w, h)
    try:
        r.render(spp=8, frame=0)
        want, st = orc.render_frame(camera.camera_uniform(w, h), w, h, frame=0, spp=8, threads=4)
        s = r.stats()
        assert int(s["rays"]) == int(st["rays"])
        assert int(s["rays_per_depth"][0]) == w * h * 8
    finally:
        r.close()


def test_chunked_frame_equals_single_chunk(cornell):
    w = h = 64
    a = _renderer(cornell, w, h)
    b = _renderer(cornell, w, h)
    try:
        b.ctx.set_max_paths(5000)   # forces several pixel chunks, one of them partial
        a.render(spp=4, frame=3)
        b.render(spp=4, frame=3)
        assert _float_eq(a.last_result(), b.last_result())
        assert _float_eq(a.accum_host(), b.accum_host())
    finally:
        a.close()
        b.close()


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_accumulation_modes(mode, cornell):
    import oracle as O
    from pt_amd import camera
    w = h = 48
    orc = _oracle(cornell)
    r = _renderer(cornell, w, h)
    try:
        cam = camera.camera_uniform(w, h)
        acc = np.zeros((w * h, 4), dtype=np.uint8 if mode == 2 else np.float32)
        for frame in range(3):
            r.render(spp=4, frame=frame, accum_mode=mode)
            res, _ = orc.render_frame(cam, w, h, frame=frame, spp=4, threads=4)
            O.accumulate(mode, frame, res, acc)
        got = r.accum_host().reshape(w * h, 4)
        if mode == 2:
            assert np.array_equal(got, acc)
        else:
            assert _float_eq(got, acc)
        if mode == 0:
            assert np.array_equal(r.resolve().reshape(-1, 4), O.resolve(acc))
            bgra = r.resolve(bgra=True).reshape(-1, 4)
            assert np.array_equal(bgra[:, [2, 1, 0, 3]], O.resolve(acc))
    finally:
        r.close()


@pytest.mark.parametrize("mode", [0, 2])
def test_frames_in_flight_keep_submission_order(mode, cornell):
    """Several frames (and several pixel chunks per frame) in flight on the library's internal
    streams give the image of strictly serial frames, bit for bit, also for the order-dependent
    8-bit running mean."""
    w = h = 256
    images = {}
    for lanes in (1, 3):
        r = _renderer(cornell, w, h)
        try:
            r.ctx.set_frames_in_flight(lanes)
            r.ctx.set_max_paths(100_000)          # three chunks per frame
            for frame in range(5):
                r.render(spp=4, frame=frame, accum_mode=mode)
            images[lanes] = r.accum_host().copy()
            assert int(r.stats()["paths"]) == 5 * w * h * 4
        finally:
            r.close()
    if mode == 2:
        assert np.array_equal(images[1], images[3])
    else:
        assert _float_eq(images[1], images[3])


@pytest.mark.parametrize("name,strip", [("dragon", (800, 500, 1056, 532)), ("cornell", (700, 520, 1212, 552)), ("cube", (832, 600, 1088, 632))])
def test_full_size_properties_1080p(name, strip, request):
    """BASELINE size (1920x1080) for the dragon, the Cornell box and the cube: properties that need no
    oracle run -- every path accounted for, one primary ray per path, monotone ray counts per bounce,
    finite non-negative radiance, sample-count linearity of the float sum -- and a strip of the frame
    against the oracle, bit for bit."""
    from pt_amd import camera
    scene = request.getfixturevalue(name)
    w, h, spp = 1920, 1080, 2
    r = _renderer(scene, w, h)
    try:
        r.render(spp=spp, frame=0)
        s = r.stats()
        assert int(s["paths"]) == w * h * spp
        assert int(s["rays_per_depth"][0]) == w * h * spp
        rp = s["rays_per_depth"][:10].astype(np.int64)
        assert np.all(np.diff(rp) <= 0)
        res = r.last_result()
        assert np.isfinite(res).all() and (res >= 0).all()
        x0, y0, x1, y1 = strip
        want, st = _oracle(scene).render_frame(camera.camera_uniform(w, h), w, h, frame=0, spp=spp, rect=strip, threads=8)
        assert int(st["hits"]) > (x1 - x0) * (y1 - y0) // 4          # the strip lies on the scene, not on the background
        assert _float_eq(res[y0:y1, x0:x1], want[y0:y1, x0:x1])
        r.render(spp=spp, frame=1)
        acc = r.accum_host()
        assert np.all(acc[..., 3] == 2.0)
    finally:
        r.close()


def test_helmet_native_texture_footprint_1080p(helmet):
    """BASELINE config 3 at the footprint of the asset: the helmet's four sampled base-colour maps at their
    native 2048 x 2048 / 1024 x 1024 (52 MiB of texels on the device; the committed fixture holds them
    reduced, they are enlarged here by a fixed rule), 1920x1080 with a camera that frames the helmet:
    path bookkeeping, and a strip across the helmet against the oracle, bit for bit."""
    from pt_amd import camera, procedural
    scene = procedural.with_native_texture_sizes(helmet)
    sizes = sorted(im.shape[0] for im in scene.tables.images if im.shape[0] > 2)
    assert sizes == [1024, 2048, 2048, 2048]
    W, H, spp = 1920, 1080, 4
    cam = camera.camera_uniform(W, H, **HELMET_CAMERA)
    r = _renderer(scene, W, H, camera=cam)
    try:
        assert sorted(int(t["width"]) for t in r.textures if t["width"] > 2) == sizes
        r.render(spp=spp, frame=0)
        s = r.stats()
        assert int(s["paths"]) == W * H * spp and int(s["rays_per_depth"][0]) == W * H * spp
        res = r.last_result()
        assert np.isfinite(res).all() and (res >= 0).all()
        strip = (640, 500, 1280, 524)
        want, st = _oracle(scene).render_frame(cam, W, H, frame=0, spp=spp, rect=strip, threads=8)
        assert int(st["tex_hits"]) > 5000
        x0, y0, x1, y1 = strip
        assert _float_eq(res[y0:y1, x0:x1], want[y0:y1, x0:x1])
        # enlarging changed what is sampled: the reduced maps give another image
        want_small, _ = _oracle(helmet).render_frame(cam, W, H, frame=0, spp=spp, rect=strip, threads=8)
        assert not _float_eq(want[y0:y1, x0:x1], want_small[y0:y1, x0:x1])
    finally:
        r.close()


def test_full_size_step_in_one_chunk_equals_pixel_chunks(dragon):
    """The benchmark's step (1080p x 256 spp = 531 M paths resident at once, slot indices beyond 2^29)
    gives, bit for bit, the frame and the ray counts of the same step traced in pixel chunks of
    2^27 paths."""
    w, h, spp = 1920, 1080, 256
    got = {}
    for name, max_paths in (("one", w * h * spp), ("chunks", 1 << 27)):
        r = _renderer(dragon, w, h)
        try:
            r.ctx.set_frames_in_flight(1)
            r.ctx.set_max_paths(max_paths)
            r.render(spp=spp, frame=5)
            s = r.stats()
            assert int(s["paths"]) == w * h * spp
            got[name] = (r.last_result().copy(), s["rays_per_depth"].astype(np.int64).copy())
        finally:
            r.close()
    assert _float_eq(got["one"][0], got["chunks"][0])
    assert np.array_equal(got["one"][1], got["chunks"][1])
    assert np.isfinite(got["one"][0]).all()


def test_errors_are_reported(cornell):
    from pt_amd import hip
    r = _renderer(cornell, 16, 16)
    try:
        with pytest.raises(hip.PtError):
            r.render(spp=70000)
        with pytest.raises(hip.PtError):
            r.render(spp=4, max_depth=64)
    finally:
        r.close()


def test_shading_follows_the_geometry_nodes_buffers():
    """closest_hit.rchit:57-60 reads vertices through the geometry node's addresses, not through what the
    BLAS was built from: pointing the nodes at a second vertex buffer (same positions, mirrored
    normals) and binding the tables again must give the image of a scene modelled that way."""
    import copy
    from pt_amd import camera, hip
    from pt_amd.scene import Scene
    path = os.path.join(os.path.dirname(__file__), "golden", "scene_cornell.npz")
    scene, mirrored = Scene.load_npz(path), Scene.load_npz(path)
    for m in mirrored.models:
        m.vertices["normal"] *= -1.0
    w = h = 48
    cam = camera.camera_uniform(w, h)
    want_plain, _ = _oracle(scene).render_frame(cam, w, h, frame=0, spp=8, threads=4)
    want_mirrored, _ = _oracle(mirrored).render_frame(cam, w, h, frame=0, spp=8, threads=4)
    assert not _float_eq(want_plain, want_mirrored)
    r = _renderer(scene, w, h)
    try:
        r.render(spp=8, frame=0)
        assert _float_eq(r.last_result(), want_plain)
        # second vertex buffer per model; the BLASes keep the first
        new_bufs = [r.ctx.upload(m.vertices) for m in mirrored.models]
        gn = copy.deepcopy(r.geometry_nodes)
        for k, inst in enumerate(scene.instances()):
            gn[k]["vertexBufferDeviceAddress"] = new_bufs[inst.model]
        hip._check(r.ctx.lib.pt_scene_set_tables(r.ctx.h, hip._p(r.materials), len(r.materials), hip._p(gn), len(gn),
                                                 hip._p(r.blas_instances), len(r.blas_instances), hip._p(r.textures), len(r.textures)))
        r.render(spp=8, frame=0)
        assert _float_eq(r.last_result(), want_mirrored)
    finally:
        r.close()


def test_trace_profile_accounts_for_every_launch(cornell):
    """With profiling on, every traversal launch is timed once: one k_primary per frame, one k_intersect
    per later bounce, and their ray counts add up to the frame statistics; results are unchanged."""
    w = h = 64
    a = _renderer(cornell, w, h)
    b = _renderer(cornell, w, h)
    try:
        b.set_profiling(True)
        for frame in range(3):
            a.render(spp=8, max_depth=6, frame=frame)
            b.render(spp=8, max_depth=6, frame=frame)
        assert _float_eq(a.accum_host(), b.accum_host())
        p, st = b.profile(), b.stats()
        assert int(p["primary_launches"]) == 3 and int(p["intersect_launches"]) == 3 * 5
        assert int(p["primary_rays"]) == 3 * w * h * 8 == int(st["rays_per_depth"][0])
        assert int(p["primary_rays"]) + int(p["intersect_rays"]) == int(st["rays"])
        assert float(p["primary_ms"]) > 0.0 and float(p["intersect_ms"]) > 0.0
        # the pairs are consumed: a second call reports nothing new
        q = b.profile()
        assert int(q["primary_launches"]) == 0 and int(q["intersect_launches"]) == 0 and int(q["intersect_rays"]) == 0
    finally:
        a.close()
        b.close()
