"""CPU suite: pins the oracle against every known answer available for this path.

The reference ships no tests or golden vectors (parity unpinned); what can be pinned is derived
from the shader text itself: TEA / LCG values computed with exact u32 arithmetic, the closed-form
primary rays of the hard-coded camera, furnace-test algebra of the bounce loop, brute force vs BVH.
"""
import hashlib
import math
import os

import numpy as np
import pytest

import oracle as O
from pt_amd import abi, camera, procedural
from pt_amd.scene import Scene

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---- shaders/sampling.glsl:4-34 ------------------------------------------------------------
@pytest.mark.parametrize("a,b,want", [
    (0, 0, 0x741c187d), (1, 0, 0x8da6b311), (0, 1, 0x70d3aef1), (1, 2, 0x0ca5550f),
    (12345, 67890, 0x51ebcd03), (131328, 0, 0xa4368b2c), (1037760, 7, 0xc908dec6),
    (0xFFFFFFFF, 0xFFFFFFFF, 0x16e50358)])
def test_tea_known_answers(a, b, want):
    assert O.tea(a, b) == want


def test_tea_matches_independent_python():
    def tea(v0, v1):
        s0 = 0
        for _ in range(16):
            s0 = (s0 + 0x9e3779b9) & 0xFFFFFFFF
            v0 = (v0 + ((((v1 << 4) + 0xa341316c) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4)) & 0xFFFFFFFF)) & 0xFFFFFFFF
            v1 = (v1 + ((((v0 << 4) + 0xad90777d) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761e)) & 0xFFFFFFFF)) & 0xFFFFFFFF
        return v0
    rng = np.random.default_rng(5)
    for a, b in rng.integers(0, 2**32, size=(200, 2)):
        assert O.tea(int(a), int(b)) == tea(int(a), int(b))


def test_lcg_and_random_known_answers():
    states, r24 = O.lcg_sequence(0x741c187d, 4)
    assert states == [0xb7d2ffb8, 0x9d013fb7, 0xedc119aa, 0xc110fd01]
    assert r24 == [13828024, 81847, 12655018, 1113345]
    got = O.random_sequence(0x741c187d, 4)
    assert got == [v / 16777216.0 for v in r24]            # exact in binary32
    assert all(0.0 <= x < 1.0 for x in O.random_sequence(12345, 1000))


def test_pinned_sincos_accuracy_and_quadrants():
    xs = np.linspace(0.0, 6.2831855, 20001, dtype=np.float32)
    err = 0.0
    for x in xs:
        s, c = O.sincos(float(x))
        err = max(err, abs(s - math.sin(float(x))), abs(c - math.cos(float(x))))
        assert abs(s * s + c * c - 1.0) < 1e-6
    assert err < 2e-7          # far inside GLSL's 2^-11 absolute error allowance for sin/cos
    assert O.sincos(0.0) == (0.0, 1.0)


def test_coordinate_system_and_hemisphere():
    rng = np.random.default_rng(1)
    for _ in range(200):
        n = rng.normal(size=3)
        n = (n / np.linalg.norm(n)).astype(np.float32)
        t = np.zeros(3, np.float32)
        b = np.zeros(3, np.float32)
        O.lib.or_create_coordinate_system(O._p(n), O._p(t), O._p(b))
        assert abs(float(t @ n)) < 1e-6 and abs(float(b @ n)) < 1e-6 and abs(float(t @ b)) < 1e-6
        assert abs(np.linalg.norm(t) - 1) < 1e-6 and abs(np.linalg.norm(b) - 1) < 1e-6
        seed = np.array([rng.integers(0, 2**32)], dtype=np.uint32)
        s0 = int(seed[0])
        d = np.zeros(3, np.float32)
        O.lib.or_sampling_hemisphere(O._p(seed), O._p(t), O._p(b), O._p(n), O._p(d))
        assert abs(np.linalg.norm(d) - 1) < 1e-5 and float(d @ n) > 0.0
        # r1 is drawn before r2 (sampling.glsl:42-43): cos(theta) = sqrt(1 - r1)
        r1, _ = O.random_sequence(s0, 2)
        assert abs(float(d @ n) - math.sqrt(1.0 - r1)) < 1e-5


# ---- shaders/ray_gen.rgen:21-27 + source/renderer.cpp:172-193 --------------------------------
def test_primary_rays_closed_form():
    for (w, h), want in (((512, 512), (-0.446689, 0.446689, -0.775202)), ((1920, 1080), (-0.664256, 0.373493, -0.647508))):
        cam = camera.camera_uniform(w, h)
        o, d = O.primary_ray(cam, w, h, 0, 0)
        assert np.array_equal(o, np.array([0, 1, 3], np.float32))
        assert np.allclose(d, want, atol=2e-6)
        # view-space direction is proportional to (dx * aspect * tan30, -dy * tan30, -1)
        for (x, y) in ((0, 0), (w - 1, 0), (w // 3, h // 5), (w - 1, h - 1)):
            _, d = O.primary_ray(cam, w, h, x, y)
            dx, dy = (x + 0.5) / w * 2 - 1, (y + 0.5) / h * 2 - 1
            v = np.array([dx * (w / h) * math.tan(math.radians(30)), -dy * math.tan(math.radians(30)), -1.0])
            assert np.allclose(d, v / np.linalg.norm(v), atol=3e-6)


def test_invert_transform():
    rng = np.random.default_rng(2)
    for _ in range(50):
        m = np.eye(4)
        m[:3, :4] = rng.normal(size=(3, 4))
        w = np.eye(4)
        w[:3, :4] = O.invert_transform(m[:3, :4].astype(np.float32)).reshape(3, 4)
        assert np.allclose(w @ np.float32(m).astype(np.float64), np.eye(4), atol=1e-4)


# ---- instance numbering (source/renderer.cpp:457-470) -----------------------------------------
def test_cornell_layout(cornell):
    inst = cornell.instances()
    assert len(inst) == 8
    model = cornell.models[0]
    assert [model.meshes[i.mesh].index_count // 3 for i in inst] == [2, 2, 2, 2, 2, 10, 10, 2]
    assert len(model.vertices) == 64 and len(cornell.tables.images) == 1
    mats = cornell.tables.material_array()
    assert np.allclose(mats[7]["emissiveFactor"], 10.0) and np.allclose(mats[7]["albedoFactor"], 1.0)
    assert np.allclose(mats[4]["albedoFactor"][:3], (0.63, 0.065, 0.05))
    orc = O.OracleScene(cornell)
    lo = np.min([orc.tlas.instance_box(i)[0] for i in range(8)], axis=0)
    hi = np.max([orc.tlas.instance_box(i)[1] for i in range(8)], axis=0)
    assert np.allclose(lo, (-1.02, 0.0, -1.04), atol=2e-3) and np.allclose(hi, (1.0, 1.99, 0.99), atol=2e-3)


# ---- BVH --------------------------------------------------------------------------------------
def _check_bvh(nodes, leaf_prims, n_prims, max_leaf):
    seen = np.zeros(n_prims, dtype=np.int32)
    inner_seen = np.zeros(len(nodes), dtype=np.int32)
    stack = [0]
    while stack:
        i = stack.pop()
        inner_seen[i] += 1
        nd = nodes[i]
        for c in (int(nd["c0"]), int(nd["c1"])):
            if c == -2**31:
                continue
            if c >= 0:
                stack.append(c)
            else:
                code = ~c
                first, cnt = code >> 3, (code & 7) + 1
                assert cnt <= max_leaf
                seen[first:first + cnt] += 1
    assert np.all(seen == 1) and np.all(inner_seen == 1)
    assert sorted(leaf_prims.tolist()) == list(range(n_prims))


@pytest.mark.parametrize("name", ["cornell", "cube", "soup", "dragon", "helmet"])
def test_bvh_is_a_partition(name, request):
    scene = request.getfixturevalue(name)
    orc = O.OracleScene(scene)
    for b in orc.blases:
        _check_bvh(b.nodes(), b.leaf_prims(), b.prim_count, 4)
        keys = b.sorted_keys()
        assert np.all(keys[:-1] <= keys[1:])
    _check_bvh(orc.tlas.nodes(), orc.tlas.leaf_instances(), orc.tlas.n, 1)


@pytest.mark.parametrize("name,n", [("cornell", 20000), ("cube", 20000), ("soup", 4000), ("dragon", 300)])
def test_bvh_traversal_equals_brute_force(name, n, request):
    scene = request.getfixturevalue(name)
    orc = O.OracleScene(scene)
    rng = np.random.default_rng(7)
    o = rng.uniform(-2.5, 3.0, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    hb, tb = orc.tlas.intersect(o, d, use_bvh=False)
    hv, tv = orc.tlas.intersect(o, d, use_bvh=True)
    assert np.array_equal(hb, hv) and np.array_equal(tb, tv)
    assert (hv["instance"] != 0xFFFFFFFF).mean() > 0.02


def test_helmet_bvh_equals_brute_force(helmet):
    """Rays aimed at the FlightHelmet (it is 0.66 units tall around the origin)."""
    orc = O.OracleScene(helmet)
    rng = np.random.default_rng(4)
    o = rng.uniform(-0.8, 0.8, size=(400, 3)).astype(np.float32)
    tgt = rng.uniform(-0.15, 0.15, size=(400, 3)).astype(np.float32) * np.array([1, 2, 1], np.float32)
    d = tgt - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    hb, tb = orc.tlas.intersect(o, d, use_bvh=False)
    hv, tv = orc.tlas.intersect(o, d, use_bvh=True)
    assert np.array_equal(hb, hv) and np.array_equal(tb, tv)
    hit = hv["instance"] != 0xFFFFFFFF
    assert hit.mean() > 0.3 and len(set(hv["instance"][hit].tolist())) >= 4


def test_exact_ties_go_to_the_lowest_instance(stacked):
    """Identical copies of one mesh: every hit ties in t across the copies; instance 0 must win
    in the BVH traversal as in the brute-force loop."""
    orc = O.OracleScene(stacked)
    rng = np.random.default_rng(9)
    o = rng.uniform(-2.0, 2.0, size=(4000, 3)).astype(np.float32)
    d = rng.normal(size=(4000, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    hb, tb = orc.tlas.intersect(o, d, use_bvh=False)
    hv, tv = orc.tlas.intersect(o, d, use_bvh=True)
    assert np.array_equal(hb, hv) and np.array_equal(tb, tv)
    hit = hv["instance"] != 0xFFFFFFFF
    assert hit.mean() > 0.1 and np.all(hv["instance"][hit] == 0)


def test_degenerate_directions_and_edges(cube):
    """Axis-aligned rays (zero direction components), rays through vertices and along edges."""
    orc = O.OracleScene(cube)
    o = np.array([[0, 0, 5], [0, 0, 5], [1, 1, 5], [1, 0, 5], [0, 0, 0], [5, 5, 5], [0, 0, 5]], np.float32)
    d = np.array([[0, 0, -1], [0, 0, 1], [0, 0, -1], [0, 0, -1], [1, 0, 0], [-1, -1, -1], [0, 0, 0]], np.float32)
    hb, tb = orc.tlas.intersect(o, d, use_bvh=False)
    hv, tv = orc.tlas.intersect(o, d, use_bvh=True)
    assert np.array_equal(hb, hv) and np.array_equal(tb, tv)
    assert hv["instance"][0] == 0 and abs(tv[0] - 4.0) < 1e-5      # front face
    assert hv["instance"][1] == 0xFFFFFFFF                          # pointing away
    assert hv["instance"][4] == 0 and abs(tv[4] - 1.0) < 1e-5      # from inside: back faces are hit
    assert hv["instance"][6] == 0xFFFFFFFF                          # zero direction never hits


# ---- furnace tests: algebra of ray_gen.rgen:52-74 ------------------------------------------------
def _closed_cube(cube, albedo, emissive):
    s = Scene()
    s.tables.images = list(cube.tables.images)
    m = abi.default_material()
    m["albedoFactor"] = (albedo, albedo, albedo, 1.0)
    m["emissiveFactor"] = (emissive,) * 3
    s.tables.materials = [m]
    # the shading normal is never flipped towards the ray (closest_hit.rchit:63), so a box that is
    # to be seen from inside needs inward-pointing vertex normals
    import copy
    model = copy.copy(cube.models[0])
    model.vertices = cube.models[0].vertices.copy()
    model.vertices["normal"] *= -1
    s.models = [model]
    return s


def test_white_furnace_is_black_and_watertight(cube):
    """Closed cube seen from inside, albedo 1, no emission: radiance is exactly zero unless a path
    escapes.  The only legitimate escape is a bounce that starts closer than tMin = 0.001 to the next
    wall (ray_gen.rgen:30), which happens in corners; anything more frequent would be a leak."""
    orc = O.OracleScene(_closed_cube(cube, 1.0, 0.0))
    w = h = 48
    cam = camera.camera_uniform(w, h, eye=(0.1, 0.2, 0.5), center=(0, 0, 0))
    res, st = orc.render_frame(cam, w, h, spp=8, threads=4)
    assert (res == 0.0).mean() > 0.97
    assert int(st["hits"]) / int(st["rays"]) > 0.9995
    leaked = res[res != 0.0]
    assert np.allclose(leaked / np.float32(0.01 * 2.5 / 8), np.rint(leaked / np.float32(0.01 * 2.5 / 8)), atol=1e-3)


def test_emissive_furnace_geometric_series(cube):
    e, rho = 0.5, 0.5
    orc = O.OracleScene(_closed_cube(cube, rho, e))
    w = h = 32
    cam = camera.camera_uniform(w, h, eye=(0.1, 0.2, 0.5), center=(0, 0, 0))
    res, _ = orc.render_frame(cam, w, h, spp=4, threads=4)
    want = 2.5 * e * sum(rho ** k for k in range(10))
    assert np.median(res) == pytest.approx(want, rel=2e-5)
    assert (np.abs(res - want) < 2e-5 * want).mean() > 0.95     # the rest lost a tail to a tMin corner escape


def test_miss_values(cube):
    """Primary miss = 0.8 * 2.5 (miss.rmiss:10-14); a camera that sees nothing gives exactly that."""
    orc = O.OracleScene(cube)
    w = h = 8
    cam = camera.camera_uniform(w, h, eye=(0, 0, 10), center=(0, 0, 20))
    res, st = orc.render_frame(cam, w, h, spp=3, threads=1)
    assert np.allclose(res, 2.0, rtol=1e-6) and int(st["rays"]) == w * h * 3


# ---- texture sampling (closest_hit.rchit:76-79) ---------------------------------------------------
def test_bilinear_repeat_sampling(cube):
    img = cube.tables.images[1]
    H, W, _ = img.shape
    rng = np.random.default_rng(3)
    for u, v in rng.uniform(-2.0, 3.0, size=(300, 2)):
        got = O.texture_bilinear(img, u, v)
        U, V = np.float32(u) * W - 0.5, np.float32(v) * H - 0.5
        i0, j0 = math.floor(U), math.floor(V)
        a, b = U - i0, V - j0
        c = lambda i, j: img[j % H, i % W, :].astype(np.float64) / 255.0  # noqa: E731
        want = (c(i0, j0) * (1 - a) + c(i0 + 1, j0) * a) * (1 - b) + (c(i0, j0 + 1) * (1 - a) + c(i0 + 1, j0 + 1) * a) * b
        assert np.allclose(got, want, atol=3e-4)
    # texel centres reproduce the texel exactly
    for (x, y) in ((0, 0), (17, 200), (W - 1, H - 1)):
        got = O.texture_bilinear(img, (x + 0.5) / W, (y + 0.5) / H)
        assert np.allclose(got, img[y, x] / 255.0, atol=1e-6)


# ---- accumulation (ray_gen.rgen:76-87) ------------------------------------------------------------
def test_accumulation_modes():
    rng = np.random.default_rng(4)
    frames = [rng.uniform(0, 1.2, size=(64, 3)).astype(np.float32) for _ in range(5)]
    s = np.zeros((64, 4), np.float32)
    m = np.zeros((64, 4), np.float32)
    q = np.zeros((64, 4), np.uint8)
    for f, r in enumerate(frames):
        O.accumulate(0, f, r, s)
        O.accumulate(1, f, r, m)
        O.accumulate(2, f, r, q)
    mean = np.mean(frames, axis=0)
    assert np.all(s[:, 3] == 5) and np.allclose(s[:, :3] / 5, mean, rtol=1e-6)
    assert np.allclose(m[:, :3], mean, rtol=1e-5) and np.all(m[:, 3] == 1)
    assert np.all(q[:, 3] == 255)
    assert np.max(np.abs(q[:, :3] / 255.0 - np.clip(mean, 0, 1))) < 0.2   # 8-bit running mean drifts: that is the literal behaviour
    out = O.resolve(s)
    assert np.array_equal(out[:, :3], np.rint(np.clip(s[:, :3] / 5, 0, 1) * 255).astype(np.uint8))


# ---- committed golden outputs of BASELINE config 1 ----------------------------------------------------
def test_cfg1_golden(cornell):
    g = np.load(os.path.join(GOLDEN, "cfg1_cornell_512_16spp.npz"))
    orc = O.OracleScene(cornell)
    W = H = 512
    cam = camera.camera_uniform(W, H)
    res, st = orc.render_frame(cam, W, H, frame=0, spp=16, threads=os.cpu_count() or 1)
    assert int(st["rays"]) == int(g["rays"])
    assert np.array_equal(res[192:320, 192:320], g["crop"])
    assert hashlib.sha256(res.tobytes()).digest() == g["sha256"].tobytes()
    # brute force (no BVH) on a strip of the same frame
    strip, _ = orc.render_frame(cam, W, H, frame=0, spp=16, rect=(200, 250, 312, 262), use_bvh=False, threads=os.cpu_count() or 1)
    assert np.array_equal(strip[250:262, 200:312], res[250:262, 200:312])
    ids = g["bounce0_ids"]
    # expected Cornell numbering: 0 floor, 1 ceiling, 2 back wall, 3 right wall, 4 left wall, 5 short box, 6 tall box, 7 light
    assert ids[440, 256, 0] == 0 and ids[96, 256, 0] in (1, 7) and ids[200, 256, 0] == 2
    assert ids[256, 80, 0] == 4 and ids[256, 440, 0] == 3 and ids[5, 5, 0] == 255
    assert ids[300, 200, 0] == 6 and ids[360, 320, 0] == 5


def test_dragon_standin_is_reproducible(dragon):
    model = dragon.models[0]
    assert dragon.triangle_count() == 134995 and len(dragon.instances()) == 2
    assert [model.meshes[i.mesh].index_count // 3 for i in dragon.instances()] == [43779, 91216]
    h = hashlib.sha256(model.vertices.tobytes() + model.indices.tobytes()).hexdigest()
    with open(os.path.join(GOLDEN, "dragon_standin.sha256")) as f:
        assert h == f.read().strip()
