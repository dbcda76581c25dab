"""Deterministic procedural scenes.

dragon_standin(): the reference's assets/dragon/DragonAttenuation.gltf needs DragonAttenuation.bin,
which is not part of the reference checkout.  This generator builds geometry of the same size
class from the facts the .gltf itself states: two root nodes under a synthetic root ->
two instances, a 43,779-triangle cloth backdrop (scale 3.5, albedo map) and a 91,216-triangle
closed dragon body (scale 0.25, +90 degrees about X), inside the accessor bounds and with the
node transforms of DragonAttenuation.gltf:33-68,205-273.  Every vertex is a closed-form function
of integer grid coordinates, so the scene is identical wherever it is generated.
"""
import numpy as np

from . import abi
from .gltf import _trs_matrix
from .scene import Mesh, Model, Node, Scene, f32


def checkerboard(size=2048, squares=16):
    """Stand-in for assets/dragon/checkerboard.jpg (2048x2048): two greys, RGBA8."""
    y, x = np.mgrid[0:size, 0:size]
    c = (((x * squares) // size + (y * squares) // size) & 1).astype(np.uint8)
    img = np.empty((size, size, 4), dtype=np.uint8)
    img[..., 0] = 60 + 170 * c
    img[..., 1] = 60 + 165 * c
    img[..., 2] = 64 + 150 * c
    img[..., 3] = 255
    return img


def _grid_indices(nu, nv, wrap_u=False):
    """Two triangles per quad of an (nu x nv)-quad grid; vertex (i, j) = j * stride + i."""
    stride = nu if wrap_u else nu + 1
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="xy")
    i = i.reshape(-1)
    j = j.reshape(-1)
    i1 = (i + 1) % nu if wrap_u else i + 1
    a = j * stride + i
    b = j * stride + i1
    c = (j + 1) * stride + i
    d = (j + 1) * stride + i1
    tri = np.stack([a, b, c, b, d, c], axis=1).reshape(-1, 3)
    return tri.astype(np.uint32)


def _vertex_normals(pos, tri):
    e1 = pos[tri[:, 1]] - pos[tri[:, 0]]
    e2 = pos[tri[:, 2]] - pos[tri[:, 0]]
    fn = np.cross(e1, e2)
    n = np.zeros_like(pos)
    for k in range(3):
        np.add.at(n, tri[:, k], fn)
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    return (n / np.where(ln > 0, ln, 1)).astype(f32)


def _cloth():
    """Backdrop sweep: floor, quarter-circle cove, back wall; 148 x 148 quads, first 43,779 triangles."""
    nu = nv = 148
    lo = np.array([-0.8216659426689148, 0.0015843699220567942, -0.46063709259033203])
    hi = np.array([0.9809443354606628, 1.0221298933029175, 1.4283748865127563])
    u = np.arange(nu + 1) / nu
    v = np.arange(nv + 1) / nv
    U, V = np.meshgrid(u, v, indexing="xy")
    x = lo[0] + (hi[0] - lo[0]) * U
    # profile along V: floor from z=hi to the cove, then up the wall
    r = 0.35
    floor_len = (hi[2] - lo[2]) - r
    wall_len = (hi[1] - lo[1]) - r
    arc_len = 0.5 * np.pi * r
    total = floor_len + arc_len + wall_len
    s = V * total
    z = np.where(s < floor_len, hi[2] - s,
                 np.where(s < floor_len + arc_len, lo[2] + r - r * np.sin((s - floor_len) / r), lo[2]))
    y = np.where(s < floor_len, lo[1],
                 np.where(s < floor_len + arc_len, lo[1] + r - r * np.cos((s - floor_len) / r), lo[1] + r + (s - floor_len - arc_len)))
    ripple = 0.004 * np.sin(37.0 * U) * np.sin(29.0 * V)
    y = np.clip(y + ripple * (s < floor_len), lo[1], hi[1])
    z = np.clip(z + ripple * (s >= floor_len + arc_len), lo[2], hi[2])
    pos = np.stack([x, y, z], axis=-1).reshape(-1, 3).astype(f32)
    tri = _grid_indices(nu, nv)[:43779]
    uv = np.stack([U * 4.0 - 1.5, V * 4.0 - 1.5], axis=-1).reshape(-1, 2).astype(f32)  # outside [0,1]: REPEAT
    return pos, _vertex_normals(pos, tri), uv, tri


def _body():
    """Closed genus-0 body: 302 x 151 lat-long sphere (91,204 triangles) displaced by fixed
    harmonics and stretched into the dragon accessor bounds, plus an octahedron and a
    tetrahedron (12 triangles) -> 91,216 triangles."""
    ns, nl = 302, 152
    lo = np.array([-7.0467000007629395, -3.1512999534606934, -9.939900398254395])
    hi = np.array([7.0467000007629395, 3.1512999534606934, 0.0])
    j = np.arange(1, nl)                       # interior rings
    th = np.pi * j / nl
    ph = 2.0 * np.pi * np.arange(ns) / ns
    TH, PH = np.meshgrid(th, ph, indexing="ij")
    rad = 1.0 + 0.18 * np.sin(5 * PH) * np.sin(3 * TH) ** 2 + 0.10 * np.cos(9 * TH) * np.sin(2 * PH + 1.0) \
        + 0.05 * np.sin(23 * PH + 4 * TH) * np.sin(TH)
    rad = rad / 1.4
    sx = np.sin(TH) * np.cos(PH) * rad
    sy = np.sin(TH) * np.sin(PH) * rad
    sz = np.cos(TH) * rad
    ring = np.stack([sx, sy, sz], axis=-1).reshape(-1, 3)
    north = np.array([[0.0, 0.0, 1.0 / 1.4]])
    south = np.array([[0.0, 0.0, -1.0 / 1.4]])
    unit = np.concatenate([ring, north, south])
    c = 0.5 * (lo + hi)
    h = 0.5 * (hi - lo)
    pos = (c + unit * h).astype(f32)
    nring = nl - 1
    tris = [_grid_indices(ns, nring - 1, wrap_u=True)]
    i = np.arange(ns, dtype=np.uint32)
    i1 = (i + 1) % ns
    npole = np.uint32(nring * ns)
    spole = np.uint32(nring * ns + 1)
    tris.append(np.stack([np.full(ns, npole, dtype=np.uint32), i1, i], axis=1))
    base = np.uint32((nring - 1) * ns)
    tris.append(np.stack([base + i, base + i1, np.full(ns, spole, dtype=np.uint32)], axis=1))
    tri = np.concatenate(tris).astype(np.uint32)[:, [0, 2, 1]]   # outward-facing winding
    assert tri.shape[0] == 2 * ns * (nl - 1)
    # two small closed solids in front of the body
    def solid(center, r, verts, faces):
        p = (np.array(center) + r * np.array(verts, dtype=np.float64)).astype(f32)
        f = np.array(faces, dtype=np.uint32)
        fn = np.cross(p[f[:, 1]] - p[f[:, 0]], p[f[:, 2]] - p[f[:, 0]])
        inward = (fn * (p[f].mean(axis=1) - np.array(center))).sum(axis=1) < 0
        f[inward] = f[inward][:, [0, 2, 1]]
        return p, f
    octa_v = [[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]]
    octa_f = [[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]]
    tet_v = [[1, 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1]]
    tet_f = [[0, 1, 2], [0, 3, 1], [0, 2, 3], [1, 3, 2]]
    extra = [solid([2.0, -2.6, -8.0], 0.4, octa_v, octa_f), solid([-2.0, -2.6, -8.0], 0.25, tet_v, tet_f)]
    for p, f in extra:
        tri = np.concatenate([tri, f + np.uint32(pos.shape[0])])
        pos = np.concatenate([pos, p])
    assert tri.shape[0] == 91216
    nrm = _vertex_normals(pos.astype(np.float64), tri)
    uv = np.zeros((pos.shape[0], 2), dtype=f32)
    return pos, nrm, uv, tri


def dragon_standin(scene: Scene = None) -> Scene:
    scene = scene or Scene()
    t = scene.tables
    cloth_tex = t.create_image(checkerboard())
    m0 = abi.default_material()
    m0["useAlbedoMap"] = 1
    m0["albedoMapIndex"] = cloth_tex
    m0["metallicFactor"] = 0.0
    m0["roughnessFactor"] = f32(0.4934999942779541)
    m1 = abi.default_material()
    m1["metallicFactor"] = 0.0
    m1["roughnessFactor"] = 0.0
    m2 = abi.default_material()
    m2["albedoFactor"] = (0.921, 0.64, 0.064, 1.0)
    m2["metallicFactor"] = 0.0
    m2["roughnessFactor"] = 0.0
    mats = [t.create_material(m) for m in (m0, m1, m2)]

    parts = [_cloth(), _body()]
    verts, inds, meshes = [], [], []
    nv = ni = 0
    for k, (pos, nrm, uv, tri) in enumerate(parts):
        v = np.zeros((pos.shape[0],), dtype=abi.VERTEX)
        v["pos"], v["normal"], v["uv"] = pos, nrm, uv
        meshes.append(Mesh(index_count=int(tri.size), first_index=ni, first_vertex=nv, material=mats[k]))
        inds.append(tri.reshape(-1) + np.uint32(nv))
        verts.append(v)
        nv += pos.shape[0]
        ni += int(tri.size)
    model = Model(vertices=np.concatenate(verts), indices=np.concatenate(inds).astype(np.uint32), name="dragon_standin")
    model.meshes = meshes
    model.materials = mats
    model.textures = [cloth_tex]
    cloth_node = {"scale": [3.5, 3.5, 3.5], "translation": [-0.15470129251480103, -0.841584324836731, -0.1703687310218811]}
    dragon_node = {"rotation": [0.7071068286895752, 0, 0, 0.7071068286895752], "scale": [0.25, 0.25, 0.25],
                   "translation": [0, -0.7306479811668396, 0]}
    model.nodes = [Node("ROOT", None, np.eye(4, dtype=f32), []),
                   Node("Cloth Backdrop", 0, _trs_matrix(cloth_node), [0]),
                   Node("Dragon", 0, _trs_matrix(dragon_node), [1])]
    scene.add_model(model)
    return scene


def triangle_soup(n_tris, seed=1, extent=1.0, size=0.1, n_instances=1) -> Scene:
    """Random small triangles in a cube (BVH / traversal parity cases), each instance gets its own
    mesh and a rigid + scale transform."""
    rng = np.random.default_rng(seed)
    scene = Scene()
    mat = scene.tables.create_material(abi.default_material())
    verts, inds, meshes, nodes = [], [], [], []
    nv = ni = 0
    per = max(1, n_tris // n_instances)
    for k in range(n_instances):
        c = rng.uniform(-extent, extent, size=(per, 1, 3))
        p = (c + rng.uniform(-size, size, size=(per, 3, 3))).reshape(-1, 3).astype(f32)
        v = np.zeros((p.shape[0],), dtype=abi.VERTEX)
        v["pos"] = p
        n = rng.normal(size=p.shape)
        v["normal"] = (n / np.linalg.norm(n, axis=1, keepdims=True)).astype(f32)
        v["uv"] = rng.uniform(0, 1, size=(p.shape[0], 2)).astype(f32)
        idx = np.arange(p.shape[0], dtype=np.uint32)
        meshes.append(Mesh(index_count=int(idx.size), first_index=ni, first_vertex=nv, material=mat))
        inds.append(idx + np.uint32(nv))
        verts.append(v)
        nv += p.shape[0]
        ni += int(idx.size)
        if k == 0:
            local = np.eye(4, dtype=f32)
        else:
            q = rng.normal(size=4)
            q /= np.linalg.norm(q)
            local = _trs_matrix({"rotation": q.tolist(), "scale": [float(rng.uniform(0.5, 1.5))] * 3,
                                 "translation": rng.uniform(-extent, extent, size=3).tolist()})
        nodes.append(Node(f"soup{k}", None, local, [k]))
    model = Model(vertices=np.concatenate(verts), indices=np.concatenate(inds).astype(np.uint32), name="soup")
    model.meshes, model.nodes, model.materials = meshes, nodes, [mat]
    scene.add_model(model)
    return scene


def stacked_copies(n_copies=6, n_tris=300, seed=5) -> Scene:
    """The same mesh instanced n_copies times with the identity transform: every hit is an exact
    tie in t between all copies, so the closest hit is decided by the (instance, primitive) order
    alone, whatever order a traversal meets the copies in."""
    rng = np.random.default_rng(seed)
    scene = Scene()
    mat = scene.tables.create_material(abi.default_material())
    c = rng.uniform(-1.0, 1.0, size=(n_tris, 1, 3))
    p = (c + rng.uniform(-0.3, 0.3, size=(n_tris, 3, 3))).reshape(-1, 3).astype(f32)
    v = np.zeros((p.shape[0],), dtype=abi.VERTEX)
    v["pos"] = p
    n = rng.normal(size=p.shape)
    v["normal"] = (n / np.linalg.norm(n, axis=1, keepdims=True)).astype(f32)
    v["uv"] = rng.uniform(0, 1, size=(p.shape[0], 2)).astype(f32)
    idx = np.arange(p.shape[0], dtype=np.uint32)
    model = Model(vertices=v, indices=idx, name="stack")
    model.meshes = [Mesh(index_count=int(idx.size), first_index=0, first_vertex=0, material=mat)]
    model.nodes = [Node(f"copy{k}", None, np.eye(4, dtype=f32), [0]) for k in range(n_copies)]
    model.materials = [mat]
    scene.add_model(model)
    return scene


# Native sizes of the FlightHelmet base-colour maps (assets/helmet/FlightHelmet_baseColor*.png: 2048 x 2048 except
# baseColor2 = LensesMat, 1024 x 1024), by the image index they have in a helmet scene imported on its own.
HELMET_NATIVE_TEXTURE_SIZES = {1: 2048, 3: 1024, 6: 2048, 9: 2048}


def upsample_texture(img, size):
    """Deterministic enlargement of a square RGBA8 texture to size x size: every texel is repeated and a
    small position-dependent pattern is added to the colour channels, so that no two neighbouring
    texels are equal (an addressing error shows) while the image stays what it was."""
    h, w = img.shape[:2]
    assert h == w and size % w == 0
    k = size // w
    big = np.repeat(np.repeat(img, k, axis=0), k, axis=1).astype(np.int32)
    y, x = np.mgrid[0:size, 0:size]
    pattern = ((x * 7 + y * 13) & 7) - 4
    big[..., :3] = np.clip(big[..., :3] + pattern[..., None], 0, 255)
    return big.astype(np.uint8)


def with_native_texture_sizes(helmet: Scene, sizes=None) -> Scene:
    """A copy of a helmet scene whose sampled base-colour maps have the asset's native sizes
    (4 x 16 MiB + 4 MiB of texels instead of the reduced fixtures)."""
    import copy
    out = copy.deepcopy(helmet)
    for idx, size in (sizes or HELMET_NATIVE_TEXTURE_SIZES).items():
        out.tables.images[idx] = upsample_texture(out.tables.images[idx], size)
    return out


# Where the composite scene puts the helmet and the dragon inside the Cornell box (world units of the box:
# x in [-1.02, 1], y in [0, 1.99], z in [-1.04, 0.99]; camera at (0, 1, 3) looking down -z).
COMPOSITE_HELMET_NODE = {"scale": [1.6, 1.6, 1.6], "translation": [0.42, 1.05, 0.35]}
COMPOSITE_DRAGON_NODE = {"scale": [0.35, 0.35, 0.35], "translation": [-0.3, 0.45, -0.2]}


def composite(cornell: Scene, helmet: Scene, dragon: Scene = None) -> Scene:
    """Cornell box + FlightHelmet + dragon in one TLAS: 8 + 5 + 2 = 15 instances, in that order, the helmet
    and the dragon hung under the root transforms above."""
    import copy
    scene = copy.deepcopy(cornell)
    scene.merge(helmet, _trs_matrix(COMPOSITE_HELMET_NODE), "helmet")
    scene.merge(dragon if dragon is not None else dragon_standin(), _trs_matrix(COMPOSITE_DRAGON_NODE), "dragon")
    return scene
