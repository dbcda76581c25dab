"""A second, independent restatement of the reference's shaders, in numpy binary64 and written from the
GLSL text alone, against which the CPU oracle is checked bounce by bounce.

The oracle (oracle/*.cpp) and the HIP kernels were written side by side with the same expressions, so a
misreading of the shaders shared by both would pass every oracle-vs-GPU comparison.  This file shares no
code and no arithmetic with either: other language, other precision, Moeller-Trumbore over all triangles
in world space instead of a two-level BVH with a watertight test in object space, numpy's sin/cos instead
of the pinned polynomials.  It follows single bounces (the ray the oracle records at depth b goes in, the
payload the shaders would produce comes out), so chaotic divergence of whole paths plays no part and the
tolerances stay at float32 rounding level.

Reference lines restated here: shaders/sampling.glsl:4-65 (TEA, LCG, Random, SamplingHemisphere,
CreateCoordinateSystem), shaders/closest_hit.rchit:33-86, shaders/miss.rmiss:8-21,
shaders/ray_gen.rgen:21-27,38-74."""
import numpy as np
import pytest

import oracle as O
from pt_amd import camera

M32 = 0xFFFFFFFF
PI = float(np.float32(3.14159265))      # `#define PI 3.14159265` is a float literal


def tea(v0, v1):                         # sampling.glsl:4-18
    s0 = 0
    for _ in range(16):
        s0 = (s0 + 0x9e3779b9) & M32
        v0 = (v0 + ((((v1 << 4) & M32) + 0xa341316c & M32) ^ ((v1 + s0) & M32) ^ (((v1 >> 5) + 0xc8013ea4) & M32))) & M32
        v1 = (v1 + ((((v0 << 4) & M32) + 0xad90777d & M32) ^ ((v0 + s0) & M32) ^ (((v0 >> 5) + 0x7e95761e) & M32))) & M32
    return v0


def rnd(seed):                           # sampling.glsl:22-34 -> (value in [0,1), new state)
    seed = (1664525 * seed + 1013904223) & M32
    return (seed & 0x00FFFFFF) / float(0x01000000), seed


def normalize(v):
    return v / np.sqrt(np.dot(v, v))


class World:
    """All triangles of a scene in world space (binary64) with what closest_hit.rchit reads for each."""

    def __init__(self, scene):
        self.scene = scene
        self.tri, self.inst, self.prim = [], [], []
        self.insts = scene.instances()
        for i, it in enumerate(self.insts):
            model = scene.models[it.model]
            mesh = model.meshes[it.mesh]
            idx = model.indices[mesh.first_index:mesh.first_index + mesh.index_count].reshape(-1, 3)
            p = model.vertices["pos"][idx].astype(np.float64)                     # n x 3 x 3, object space
            m = it.world.astype(np.float64)
            self.tri.append(p @ m[:3, :3].T + m[:3, 3])
            self.inst += [i] * len(idx)
            self.prim += list(range(len(idx)))
        self.tri = np.concatenate(self.tri)
        self.inst, self.prim = np.array(self.inst), np.array(self.prim)

    def closest(self, o, d, tmin=0.001, tmax=10000.0):
        """Moeller-Trumbore against every triangle, both faces: (index of the closest, t, u, v, gap to the runner-up)."""
        v0, e1, e2 = self.tri[:, 0], self.tri[:, 1] - self.tri[:, 0], self.tri[:, 2] - self.tri[:, 0]
        p = np.cross(d, e2)
        det = np.einsum("ij,ij->i", e1, p)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / det
            s = o - v0
            u = np.einsum("ij,ij->i", s, p) * inv
            q = np.cross(s, e1)
            v = np.einsum("ij,j->i", q, d) * inv
            t = np.einsum("ij,ij->i", e2, q) * inv
        ok = (det != 0) & (u >= 0) & (v >= 0) & (u + v <= 1) & (t > tmin) & (t < tmax)
        if not ok.any():
            return None
        tt = np.where(ok, t, np.inf)
        k = int(np.argmin(tt))
        rest = np.delete(tt, k)
        return k, tt[k], u[k], v[k], (rest.min() - tt[k]) if len(rest) else np.inf

    def shade(self, inst, prim, u, v, seed):
        """closest_hit.rchit main(): -> (next origin, next direction, hitValue, weight, seed after)."""
        it = self.insts[inst]
        model = self.scene.models[it.model]
        mesh = model.meshes[it.mesh]
        idx = model.indices[mesh.first_index + 3 * prim:mesh.first_index + 3 * prim + 3]
        vert = model.vertices[idx]
        b = np.array([1.0 - u - v, u, v])
        pos = (vert["pos"].astype(np.float64) * b[:, None]).sum(0)                 # :47-49
        nrm = normalize((vert["normal"].astype(np.float64) * b[:, None]).sum(0))
        uv = (vert["uv"].astype(np.float64) * b[:, None]).sum(0)
        m = it.world.astype(np.float64)
        wpos = m[:3, :3] @ pos + m[:3, 3]                                         # :62
        w2o = np.linalg.inv(m)[:3, :3]
        wn = normalize(nrm @ w2o)                                                 # :63 row vector times matrix
        if abs(wn[0]) > abs(wn[1]):                                               # sampling.glsl:53-65
            nt = np.array([wn[2], 0.0, -wn[0]]) / np.sqrt(wn[0] * wn[0] + wn[2] * wn[2])
        else:
            nt = np.array([0.0, -wn[2], wn[1]]) / np.sqrt(wn[1] * wn[1] + wn[2] * wn[2])
        nb = np.cross(wn, nt)
        r1, seed = rnd(seed)                                                      # sampling.glsl:42-43: r1 first
        r2, seed = rnd(seed)
        sq = np.sqrt(r1)
        local = np.array([np.cos(2 * PI * r2) * sq, np.sin(2 * PI * r2) * sq, np.sqrt(1.0 - r1)])
        d = local[0] * nt + local[1] * nb + local[2] * wn
        cos_theta = np.dot(d, wn)
        pdf = cos_theta / PI
        mat = self.scene.tables.materials[mesh.material]
        albedo = mat["albedoFactor"].astype(np.float64)[:3].copy()
        if mat["useAlbedoMap"]:
            albedo *= bilinear(self.scene.tables.images[int(mat["albedoMapIndex"])], uv[0], uv[1])
        weight = (albedo / PI) * cos_theta / pdf                                  # :80-85
        return wpos, d, mat["emissiveFactor"].astype(np.float64), weight, seed


def bilinear(img, u, v):
    """VkSampler LINEAR / REPEAT / unnormalizedCoordinates = false on an R8G8B8A8_UNORM image, level 0."""
    h, w = img.shape[:2]
    x, y = u * w - 0.5, v * h - 0.5
    x0, y0 = int(np.floor(x)), int(np.floor(y))
    a, b = x - x0, y - y0
    t = lambda i, j: img[j % h, i % w, :3].astype(np.float64) / 255.0
    return (t(x0, y0) * (1 - a) + t(x0 + 1, y0) * a) * (1 - b) + (t(x0, y0 + 1) * (1 - a) + t(x0 + 1, y0 + 1) * a) * b


def primary_ray(cam, W, H, px, py):      # ray_gen.rgen:21-27; camera = viewInverse then projInverse, column-major
    vi = np.asarray(cam[:16], np.float64).reshape(4, 4).T
    pi = np.asarray(cam[16:], np.float64).reshape(4, 4).T
    dx, dy = (px + 0.5) / W * 2 - 1, (py + 0.5) / H * 2 - 1
    origin = vi @ np.array([0, 0, 0, 1.0])
    target = pi @ np.array([dx, dy, 1, 1.0])
    direction = vi @ np.append(normalize(target[:3]), 0.0)
    return origin[:3], direction[:3]


@pytest.mark.parametrize("name,w,h,pixels", [("cornell", 64, 64, 60), ("cube", 48, 40, 240), ("helmet", 64, 64, 600)])
def test_oracle_follows_the_shaders_bounce_by_bounce(name, w, h, pixels, request):
    scene = request.getfixturevalue(name)
    orc = O.OracleScene(scene)
    world = World(scene)
    # the reference's camera frames the 0.3-unit helmet tiny: a closer one, so that most pixels hit its five textured instances
    cam = camera.camera_uniform(w, h, eye=(0.15, 0.1, 0.75), center=(0.0, 0.0, 0.0), fov_deg=50.0) if name == "helmet" else camera.camera_uniform(w, h)
    rng = np.random.default_rng(17)
    skipped = hits = misses = textured = 0
    for _ in range(pixels):
        px, py, s = int(rng.integers(0, w)), int(rng.integers(0, h)), int(rng.integers(0, 64))
        bounces, radiance = orc.trace_path(cam, w, h, px, py, s, max_depth=10)
        o0, d0 = primary_ray(cam, w, h, px, py)
        assert np.allclose(bounces[0]["origin"], o0, atol=1e-6) and np.allclose(bounces[0]["dir"], d0, atol=1e-6)
        seed = tea(py * w + px, s)                                                 # the deterministic stand-in for ray_gen.rgen:40
        hit_value, weight_product = np.zeros(3), np.ones(3)
        for depth, b in enumerate(bounces):
            o, d = b["origin"].astype(np.float64), b["dir"].astype(np.float64)
            got = world.closest(o, d)
            miss = int(b["hit"]["instance"]) == M32
            if got is not None and (got[4] < 1e-5 * max(1.0, got[1]) or min(got[2], got[3], 1 - got[2] - got[3]) < 1e-5):
                skipped += 1            # two surfaces within rounding of each other, or an edge: either answer is right
                break
            if miss:
                assert got is None, (px, py, s, depth)
                value = np.full(3, 0.8 if depth == 0 else 0.01)                    # miss.rmiss:10-18
                assert np.allclose(b["hit_value"], value, rtol=1e-6)
                hit_value += value * weight_product                                # ray_gen.rgen:67
                assert depth == len(bounces) - 1                                   # depth = 100 ends the loop
                misses += 1
                break
            assert got is not None, (px, py, s, depth)
            k, t, u, v, _ = got
            assert (int(world.inst[k]), int(world.prim[k])) == (int(b["hit"]["instance"]), int(b["hit"]["primitive"])), (px, py, s, depth)
            assert abs(t - b["t"]) < 1e-4 * max(1.0, t) and abs(u - b["hit"]["u"]) < 2e-4 and abs(v - b["hit"]["v"]) < 2e-4
            # the shaders from the oracle's own hit attributes, so that only the shading arithmetic is compared
            wpos, nd, emissive, weight, seed = world.shade(int(b["hit"]["instance"]), int(b["hit"]["primitive"]), float(b["hit"]["u"]), float(b["hit"]["v"]), seed)
            assert seed == int(b["seed_after"])
            assert np.allclose(b["hit_value"], emissive, rtol=1e-6, atol=1e-7)
            assert np.allclose(b["weight"], weight, rtol=2e-5, atol=1e-6)
            if scene.tables.materials[scene.models[world.insts[int(b["hit"]["instance"])].model].meshes[world.insts[int(b["hit"]["instance"])].mesh].material]["useAlbedoMap"]:
                textured += 1
            if depth + 1 < len(bounces):
                nxt = bounces[depth + 1]
                assert np.allclose(nxt["origin"], wpos, rtol=1e-5, atol=2e-6), (px, py, s, depth)
                assert np.allclose(nxt["dir"], nd, rtol=0, atol=2e-5), (px, py, s, depth)
            hit_value += emissive * weight_product
            weight_product = weight_product * weight                                # ray_gen.rgen:68
            hits += 1
        else:
            assert len(bounces) == 10                                              # ray_gen.rgen:52
        # the path's radiance is the series of ray_gen.rgen:67-68 (checked when the whole path was followed)
        if depth == len(bounces) - 1 and (miss or len(bounces) == 10):
            assert np.allclose(radiance, hit_value, rtol=1e-4, atol=1e-6), (px, py, s)
    assert hits > 40 and misses > 5 and skipped < 0.2 * pixels
    if name in ("cube", "helmet"):
        assert textured > 20
