"""Edge cases of the hot path on the GPU against the oracle: degenerate rays, the smallest and the
flattest acceleration structures, degenerate triangles, and the composite scene at 3840x2160."""
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


def _composite(cornell):
    """Cornell box + the dragon stand-in + a triangle soup inside it: 8 + 2 + 5 = 15 instances."""
    import copy
    from pt_amd import procedural
    from pt_amd.gltf import _trs_matrix
    scene = copy.deepcopy(cornell)
    procedural.dragon_standin(scene)
    scene.models[-1].nodes[0].local_matrix = _trs_matrix({"scale": [0.35, 0.35, 0.35], "translation": [-0.3, 0.45, -0.2]})
    soup = procedural.triangle_soup(3000, seed=9, extent=0.3, size=0.05, n_instances=5)
    m = soup.models[0]
    base = len(scene.tables.materials)
    scene.tables.materials.extend(soup.tables.materials)
    for mesh in m.meshes:
        mesh.material += base
    from pt_amd.scene import Node, f32
    root = Node("soup_root", None, _trs_matrix({"scale": [0.5, 0.5, 0.5], "translation": [0.4, 1.2, 0.1]}), [])
    m.nodes = [root] + [Node(n.name, 0, n.local_matrix, n.meshes) for n in m.nodes]
    scene.add_model(m)
    return scene


def test_composite_scene_at_2160p(cornell):
    """15 instances with nested transforms at 3840x2160: bounce-0 ids of a sample of primary rays
    equal the oracle's, every path is accounted for, two chunked frames in flight included."""
    import oracle as O
    from pt_amd import camera, hip
    scene = _composite(cornell)
    assert len(scene.instances()) == 15
    W, H = 3840, 2160
    orc = O.OracleScene(scene)
    r = hip.Renderer(scene, W, H)
    try:
        cam = camera.camera_uniform(W, H)
        rng = np.random.default_rng(1)
        px = rng.integers(0, W, size=20000); py = rng.integers(0, H, size=20000)
        o = np.zeros((len(px), 3), np.float32); d = np.zeros((len(px), 3), np.float32)
        for k in range(len(px)):
            o[k], d[k] = O.primary_ray(cam, W, H, int(px[k]), int(py[k]))
        hg, tg = r.intersect(o, d)
        ho, to = orc.tlas.intersect(o, d)
        assert np.array_equal(hg, ho) and _float_eq(tg, to)
        hb, tb = orc.tlas.intersect(o[:300], d[:300], use_bvh=False)
        assert np.array_equal(hb, ho[:300]) and _float_eq(tb, to[:300])
        assert len(set(ho["instance"].tolist()) - {0xFFFFFFFF}) >= 10
        r.ctx.set_max_paths(5_000_000)
        for frame in range(2):
            r.render(spp=2, frame=frame)
        s = r.stats()
        assert int(s["paths"]) == 2 * W * H * 2 and int(s["rays_per_depth"][0]) == 2 * W * H * 2
        res = r.last_result()
        assert np.isfinite(res).all() and (res >= 0).all()
        # a strip of the second frame against the oracle
        want, _ = orc.render_frame(cam, W, H, frame=1, spp=2, rect=(1800, 1000, 2056, 1064), threads=8)
        assert _float_eq(res[1000:1064, 1800:2056], want[1000:1064, 1800:2056])
    finally:
        r.close()
