"""Edge cases of the hot path on the GPU against the oracle: degenerate rays, the smallest and the
flattest acceleration structures, degenerate triangles, and the composite scene (Cornell box, helmet,
dragon) at 3840x2160."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _float_eq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


def _scene_from_triangles(pos, tri, transforms=None):
    from pt_amd import abi
    from pt_amd.scene import Mesh, Model, Node, Scene, f32
    scene = Scene()
    mat = scene.tables.create_material(abi.default_material())
    v = np.zeros((len(pos),), dtype=abi.VERTEX)
    v["pos"] = np.asarray(pos, np.float32)
    v["normal"] = (0, 0, 1)
    model = Model(vertices=v, indices=np.asarray(tri, np.uint32).reshape(-1), name="edge")
    model.meshes = [Mesh(index_count=int(np.asarray(tri).size), first_index=0, first_vertex=0, material=mat)]
    transforms = transforms or [np.eye(4, dtype=f32)]
    model.nodes = [Node(f"n{k}", None, np.asarray(m, f32), [0]) for k, m in enumerate(transforms)]
    model.materials = [mat]
    scene.add_model(model)
    return scene


def _compare(scene, o, d, schedule, monkeypatch):
    import oracle as O
    from pt_amd import hip
    monkeypatch.setenv("PT_HIP_TRAVERSAL", schedule)
    orc = O.OracleScene(scene)
    r = hip.Renderer(scene, 8, 8)
    try:
        hg, tg = r.intersect(o, d)
        ho, to = orc.tlas.intersect(o, d)
        assert np.array_equal(hg, ho) and _float_eq(tg, to)
        hb, tb = orc.tlas.intersect(o[:4000], d[:4000], use_bvh=False)      # the oracle's own brute force
        assert np.array_equal(hb, ho[:4000]) and _float_eq(tb, to[:4000])
        return ho, to
    finally:
        r.close()


@pytest.mark.parametrize("schedule", ["plain", "sched"])
def test_degenerate_rays(cube, schedule, monkeypatch):
    """Zero direction components (the slab reciprocal is clamped), a zero direction, rays through
    vertices and along edges, origins on faces and inside, and directions of tiny or huge length."""
    rng = np.random.default_rng(3)
    o = [[0, 0, 5], [0, 0, 5], [1, 1, 5], [1, 0, 5], [0, 0, 0], [5, 5, 5], [0, 0, 5], [1, 0.3, 0.2], [0, 1, 0], [0, 0, 5], [0, 0, 5]]
    d = [[0, 0, -1], [0, 0, 1], [0, 0, -1], [0, 0, -1], [1, 0, 0], [-1, -1, -1], [0, 0, 0], [-1, 0, 0], [0, -1, 0], [0, 0, -1e-30], [0, 0, -1e30]]
    # axis-aligned rays from a grid of origins, each axis and sign
    g = np.linspace(-1.0, 1.0, 9, dtype=np.float32)
    for axis in range(3):
        for sign in (-1.0, 1.0):
            for a in g:
                for b in g:
                    p = [a, b]
                    p.insert(axis, -3.0 * sign)
                    q = [0.0, 0.0]
                    q.insert(axis, sign)
                    o.append(p); d.append(q)
    # random origins with one or two direction components forced to +-0
    for k in range(3000):
        v = rng.normal(size=3)
        z = rng.integers(0, 3)
        v[z] = 0.0 if k % 2 else -0.0
        if k % 3 == 0:
            v[(z + 1) % 3] = 0.0
        o.append(rng.uniform(-2, 2, size=3)); d.append(v)
    ho, to = _compare(cube, np.array(o, np.float32), np.array(d, np.float32), schedule, monkeypatch)
    assert ho["instance"][0] == 0 and ho["instance"][1] == 0xFFFFFFFF and ho["instance"][6] == 0xFFFFFFFF
    assert (ho["instance"] != 0xFFFFFFFF).sum() > 500


@pytest.mark.parametrize("schedule", ["plain", "sched"])
def test_direction_lengths_beyond_the_short_reciprocal(cube, dragon, schedule, monkeypatch):
    """The kernels take 1/x by the hardware approximation and one correction while every operand of the wave lies in
    2^-100 .. 2^100, and by the IEEE division otherwise.  Directions of length 2^105, 2^120 and 2^-95 (whose
    components fall below the 2^-80 clamp of the slab reciprocal) mixed into waves of ordinary rays take both
    forms, with an interval wide enough for their hit distances: ids, distances and barycentrics as the oracle's."""
    import oracle as O
    from pt_amd import hip
    monkeypatch.setenv("PT_HIP_TRAVERSAL", schedule)
    rng = np.random.default_rng(41)
    for scene, ext in ((cube, 2.0), (dragon, 1.5)):
        n = 6000
        o = rng.uniform(-ext, ext, size=(n, 3)).astype(np.float32)
        o[:, 1] += 1.0 if scene is dragon else 0.0
        tgt = rng.uniform(-0.5, 0.5, size=(n, 3)).astype(np.float32)
        tgt[:, 1] += 0.6 if scene is dragon else 0.0
        d = tgt - o
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        k = np.arange(n)
        d[k % 7 == 1] *= np.float32(2.0 ** 105)
        d[k % 11 == 2] *= np.float32(2.0 ** -95)
        d[(k % 13 == 3) & (k % 7 != 1)] *= np.float32(2.0 ** 120)
        orc = O.OracleScene(scene)
        r = hip.Renderer(scene, 8, 8)
        try:
            hg, tg = r.intersect(o, d, tmin=1e-37, tmax=3e38)
            ho, to = orc.tlas.intersect(o, d, tmin=1e-37, tmax=3e38)
        finally:
            r.close()
        assert np.array_equal(hg, ho) and _float_eq(tg, to)
        scaled = (k % 7 == 1) | (k % 11 == 2) | (k % 13 == 3)
        assert (ho["instance"][scaled] != 0xFFFFFFFF).sum() > 100 and (ho["instance"][~scaled] != 0xFFFFFFFF).sum() > 100


@pytest.mark.parametrize("schedule", ["plain", "sched"])
def test_single_triangle_and_degenerate_triangles(schedule, monkeypatch):
    """A one-triangle BLAS (no inner node), then a mesh mixing proper, zero-area and repeated
    triangles (all keys equal, ties in the radix tree), instanced three times with scale and shear."""
    rng = np.random.default_rng(5)
    o = rng.uniform(-2, 2, size=(20000, 3)).astype(np.float32)
    tgt = rng.uniform(-0.6, 0.6, size=(20000, 3)).astype(np.float32)
    d = (tgt - o).astype(np.float32)
    one = _scene_from_triangles([[0, 0, 0], [1, 0, 0], [0, 1, 0]], [[0, 1, 2]])
    ho, _ = _compare(one, o, d, schedule, monkeypatch)
    assert (ho["instance"] == 0).sum() > 100
    pos = [[0, 0, 0], [1, 0, 0], [0, 1, 0], [0.5, 0.5, 0.5], [-1, -1, 0.2], [1, -1, 0.2], [0, 1, 0.2]]
    tri = [[0, 1, 2], [0, 0, 0], [3, 3, 3], [0, 1, 1], [4, 5, 6], [4, 5, 6], [4, 5, 6], [0, 1, 2], [1, 2, 3]]
    m1 = np.eye(4, dtype=np.float32)
    m2 = np.array([[0.5, 0.2, 0, 0.3], [0, 1.5, 0, -0.2], [0.1, 0, -2.0, 0.1], [0, 0, 0, 1]], np.float32)
    m3 = np.array([[1e-3, 0, 0, 0], [0, 1e-3, 0, 0], [0, 0, 1e-3, 1.0], [0, 0, 0, 1]], np.float32)
    mixed = _scene_from_triangles(pos, tri, [m1, m2, m3])
    ho, _ = _compare(mixed, o, d, schedule, monkeypatch)
    assert (ho["instance"] != 0xFFFFFFFF).sum() > 500
    # repeated triangles tie exactly: the lowest primitive index wins
    hit45 = ho[(ho["instance"] == 0) & (ho["primitive"] >= 4) & (ho["primitive"] <= 6)]
    assert len(hit45) and np.all(hit45["primitive"] == 4)


def _primary_rays(O, cam, W, H, px, py):
    o = np.zeros((len(px), 3), np.float32); d = np.zeros((len(px), 3), np.float32)
    for k in range(len(px)):
        o[k], d[k] = O.primary_ray(cam, W, H, int(px[k]), int(py[k]))
    return o, d


def test_composite_scene_at_2160p(cornell, helmet, dragon):
    """BASELINE config 5's scene (Cornell box + FlightHelmet + dragon stand-in: 8 + 5 + 2 = 15 instances under
    nested transforms, the helmet's base-colour maps bound) at 3840x2160: hit records of primary rays over
    the whole image and over the helmet equal the oracle's and the brute force's, textured helmet
    instances are among them, every path of two chunked frames in flight is accounted for, and a strip
    of the second frame that contains helmet pixels equals the oracle's bit for bit."""
    import oracle as O
    from pt_amd import camera, hip, procedural
    scene = procedural.composite(cornell, helmet, dragon)
    insts = scene.instances()
    assert len(insts) == 15 and scene.triangle_count() == 32 + 94722 + 134995
    mats = scene.tables.material_array()
    textured = {i for i, it in enumerate(insts) if mats[scene.models[it.model].meshes[it.mesh].material]["useAlbedoMap"]}
    helmet_textured = textured & set(range(8, 13))
    assert len(helmet_textured) == 4 and 13 in textured
    W, H = 3840, 2160
    orc = O.OracleScene(scene)
    r = hip.Renderer(scene, W, H)
    try:
        cam = camera.camera_uniform(W, H)
        rng = np.random.default_rng(1)
        px = np.concatenate([rng.integers(0, W, size=20000), rng.integers(2064, 2432, size=8000)])
        py = np.concatenate([rng.integers(0, H, size=20000), rng.integers(880, 1600, size=8000)])
        o, d = _primary_rays(O, cam, W, H, px, py)
        hg, tg = r.intersect(o, d)
        ho, to = orc.tlas.intersect(o, d)
        assert np.array_equal(hg, ho) and _float_eq(tg, to)
        sub = np.concatenate([np.arange(150), 20000 + np.arange(150)])
        hb, tb = orc.tlas.intersect(o[sub], d[sub], use_bvh=False)
        assert np.array_equal(hb, ho[sub]) and _float_eq(tb, to[sub])
        seen = set(ho["instance"].tolist()) - {0xFFFFFFFF}
        assert len(seen) >= 13 and helmet_textured <= seen
        assert np.isin(ho["instance"], list(helmet_textured)).sum() > 500
        r.ctx.set_max_paths(5_000_000)
        for frame in range(2):
            r.render(spp=2, frame=frame)
        s = r.stats()
        assert int(s["paths"]) == 2 * W * H * 2 and int(s["rays_per_depth"][0]) == 2 * W * H * 2
        res = r.last_result()
        assert np.isfinite(res).all() and (res >= 0).all()
        # a strip of the second frame across the helmet against the oracle; its paths do hit textured surfaces
        x0, y0, x1, y1 = 2080, 1000, 2400, 1048
        want, st = orc.render_frame(cam, W, H, frame=1, spp=2, rect=(x0, y0, x1, y1), threads=8)
        assert int(st["tex_hits"]) > 2000
        assert _float_eq(res[y0:y1, x0:x1], want[y0:y1, x0:x1])
        # the helmet's pixels are not grey: its maps really are sampled
        oh, dh = _primary_rays(O, cam, W, H, np.tile(np.arange(x0, x1), 2), np.repeat([y0, y1 - 1], x1 - x0))
        on_helmet = np.isin(orc.tlas.intersect(oh, dh)[0]["instance"], list(helmet_textured)).reshape(2, x1 - x0)
        assert on_helmet.sum() > 50
        rows = res[[y0, y1 - 1], x0:x1][on_helmet]
        assert np.mean(np.abs(rows[:, 0] - rows[:, 2])) > 1e-3
    finally:
        r.close()


@pytest.mark.parametrize("name,limit", [("cube", "0"), ("dragon", "0"), ("cornell", "8")])
def test_rays_starting_with_the_instances_give_the_same_image(name, limit, request, monkeypatch):
    """Under a TLAS of a few instances a ray starts with the instance leaves instead of the root node
    (PT_HIP_ROOT_ITEMS: largest such TLAS, default 2).  The closest hit does not depend on that: a frame with the
    other setting (root step for the one- and two-instance scenes, instance start for the eight of the Cornell box)
    is the same bit for bit, for all three traversal schedules."""
    from pt_amd import hip
    scene = request.getfixturevalue(name)

    def frame():
        r = hip.Renderer(scene, 96, 64)
        try:
            r.render(spp=8, max_depth=6, frame=3)
            img = r.accum_host().copy()
            rays = int(r.stats()["rays"])
        finally:
            r.close()
        return img, rays

    for schedule in ("auto", "sched"):
        monkeypatch.setenv("PT_HIP_TRAVERSAL", schedule)
        monkeypatch.delenv("PT_HIP_ROOT_ITEMS", raising=False)
        ref_img, ref_rays = frame()
        monkeypatch.setenv("PT_HIP_ROOT_ITEMS", limit)
        img, rays = frame()
        assert rays == ref_rays and np.array_equal(img.view(np.uint32), ref_img.view(np.uint32)), schedule


@pytest.mark.parametrize("name", ["cube", "cornell"])
def test_rays_and_cameras_that_are_not_finite_hit_nothing(name, request, monkeypatch):
    """NaN and infinite origins or directions: every comparison of the box steps fails or yields NaN, which no
    child's entry distance survives, so such a ray ends without a hit in every schedule, as in the oracle; a frame
    from a camera of NaNs is traced to its end (bounce 0 walks the float-plane nodes, whose empty children are
    (+inf, -inf) boxes) and holds the value of a first-bounce miss in every pixel."""
    from pt_amd import hip
    scene = request.getfixturevalue(name)
    nan, inf = np.nan, np.inf
    o = np.array([[nan, 0, 5], [0, 0, 5], [inf, 0, 5], [0, 0, 5], [0, 0, 5], [nan, nan, nan], [0, -inf, 0], [0, 0, 5]], np.float32)
    d = np.array([[0, 0, -1], [nan, 0, -1], [0, 0, -1], [0, inf, -1], [nan, nan, nan], [0, 0, -1], [0, 1, 0], [-inf, inf, -inf]], np.float32)
    for schedule in ("plain", "sched"):
        ho, _ = _compare(scene, o, d, schedule, monkeypatch)
        assert (ho["instance"] == 0xFFFFFFFF).all(), schedule
    monkeypatch.setenv("PT_HIP_TRAVERSAL", "auto")
    r = hip.Renderer(scene, 32, 16, camera=np.full(32, np.nan, np.float32))
    try:
        r.render(spp=4, max_depth=4, frame=0)
        img = r.last_result()
        assert int(r.stats()["rays"]) == 32 * 16 * 4          # every path ends at its first ray
        assert np.array_equal(img, np.full_like(img, img[0, 0, 0])) and np.isfinite(img).all() and img[0, 0, 0] > 0
    finally:
        r.close()
