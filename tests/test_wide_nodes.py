"""Host-side construction of the 4-wide quantised traversal nodes, checked without a GPU against
the binary BVH the oracle builds (the GPU builder's binary nodes are bit-identical to it).

A wide tree is valid when (1) it references every triangle position exactly once and every wide
node exactly once, and (2) each entry's dequantised box contains the padded boxes of all triangles
below it; with the monotone slab test (2) makes every ray reach every triangle it could hit, so
the closest hit cannot depend on the layout."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle as O
from pt_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "path-tracer_amd", "lib", "libpt_wide_host.so")

WIDE4 = abi.WIDE_NODE
EMPTY = -2**31


@pytest.fixture(scope="module")
def wide_lib():
    if not os.path.exists(LIB):
        pytest.fail("libpt_wide_host.so is not built (make -C path-tracer_amd/csrc)")
    lib = C.CDLL(LIB)
    lib.pt_wide_collapse_host.restype = C.c_uint32
    lib.pt_wide_collapse_host.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.pt_sah_build_host_c.restype = C.c_uint32
    lib.pt_sah_build_host_c.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pt_wide_candidates_host.restype = C.c_uint32
    lib.pt_wide_candidates_host.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_uint32]
    return lib


def collapse(lib, nodes, tri_base=0):
    assert WIDE4.itemsize == 64
    nodes = np.ascontiguousarray(nodes)
    out = np.zeros(len(nodes) + 1, dtype=WIDE4)
    depth = C.c_uint32(0)
    n = lib.pt_wide_collapse_host(nodes.ctypes.data, len(nodes), tri_base, out.ctypes.data, len(out), C.byref(depth))
    assert n > 0
    return out[:n].copy(), depth.value


def entry_box(w, e):
    lo, hi = np.zeros(3, np.float32), np.zeros(3, np.float32)
    for k in range(3):
        scale = w[("scale_x", "scale_y", "scale_z")[k]]
        m, _ = np.frexp(np.float64(scale))
        assert m == 0.5 and scale > 0          # a power of two
        ql = (int(w["q"][k]) >> (8 * e)) & 255
        qh = (int(w["q"][3 + k]) >> (8 * e)) & 255
        # q * scale + origin is exact in binary64, so one rounding to binary32 equals the fma
        lo[k] = np.float32(np.float64(ql) * np.float64(scale) + np.float64(w["origin"][k]))
        hi[k] = np.float32(np.float64(qh) * np.float64(scale) + np.float64(w["origin"][k]))
    return lo, hi


def padded_prim_boxes(blas):
    """Boxes of the triangles in leaf order, padded by the BLAS's delta, in binary32."""
    v = blas.vertices["pos"] if "pos" in blas.vertices.dtype.names else None
    assert v is not None
    idx = blas.indices[:3 * blas.prim_count].reshape(-1, 3)[blas.leaf_prims()]
    p = v[idx]                                   # [n, 3 corners, 3]
    d = np.float32(blas.delta())
    return (p.min(axis=1) - d).astype(np.float32), (p.max(axis=1) + d).astype(np.float32)


def check_wide(wide, depth, prim_lo, prim_hi, n_prims, tri_base=0):
    seen = np.zeros(n_prims, np.int32)
    node_seen = np.zeros(len(wide), np.int32)
    node_seen[0] = 1
    max_depth = 0

    def visit(i, d):
        nonlocal max_depth
        max_depth = max(max_depth, d)
        w = wide[i]
        lo_all = np.full(3, np.inf, np.float32)
        hi_all = np.full(3, -np.inf, np.float32)
        n_entries = 0
        for e in range(4):
            c = int(w["child"][e])
            if c == EMPTY:
                continue
            n_entries += 1
            if c >= 0:
                node_seen[c] += 1
                lo, hi = visit(c, d + 1)
            else:
                code = (~c) & 0xFFFFFFFF
                first, cnt = (code >> 3) - tri_base, (code & 7) + 1
                seen[first:first + cnt] += 1
                lo, hi = prim_lo[first:first + cnt].min(axis=0), prim_hi[first:first + cnt].max(axis=0)
            qlo, qhi = entry_box(w, e)
            assert np.all(qlo <= lo) and np.all(qhi >= hi), (i, e, qlo, lo, qhi, hi)
            lo_all, hi_all = np.minimum(lo_all, lo), np.maximum(hi_all, hi)
        assert n_entries >= 1
        return lo_all, hi_all

    visit(0, 1)
    assert np.all(seen == 1) and np.all(node_seen == 1)
    assert max_depth == depth


@pytest.mark.parametrize("name", ["cornell", "cube", "soup", "dragon"])
def test_wide_tree_is_a_conservative_partition(name, request, wide_lib):
    scene = request.getfixturevalue(name)
    orc = O.OracleScene(scene)
    done = set()
    for b in orc.blases:
        key = (b.vertices.ctypes.data, b.indices.ctypes.data, b.prim_count)
        if key in done:
            continue
        done.add(key)
        wide, depth = collapse(wide_lib, b.nodes())
        lo, hi = padded_prim_boxes(b)
        check_wide(wide, depth, lo, hi, b.prim_count)
        # fuller nodes than a binary tree: fewer wide nodes than binary nodes unless the tree is tiny
        assert len(wide) <= max(1, len(b.nodes()))
    # TLAS: leaves are instances, boxes are the instance world boxes
    leaf_inst = orc.tlas.leaf_instances()
    boxes = [orc.tlas.instance_box(int(i)) for i in leaf_inst]
    lo = np.stack([b[0] for b in boxes]); hi = np.stack([b[1] for b in boxes])
    wide, depth = collapse(wide_lib, orc.tlas.nodes())
    check_wide(wide, depth, lo, hi, orc.tlas.n)


def test_leaf_references_are_rebased(cube, wide_lib):
    orc = O.OracleScene(cube)
    b = orc.blases[0]
    w0, _ = collapse(wide_lib, b.nodes(), 0)
    w1, _ = collapse(wide_lib, b.nodes(), 1000)
    for a, c in zip(w0["child"].ravel(), w1["child"].ravel()):
        if a < 0 and a != EMPTY:
            ca, cc = (~int(a)) & 0xFFFFFFFF, (~int(c)) & 0xFFFFFFFF
            assert (cc >> 3) == (ca >> 3) + 1000 and (cc & 7) == (ca & 7)
        else:
            assert a == c


@pytest.mark.parametrize("name,n", [("cube", 3000), ("soup", 1500), ("dragon", 300)])
def test_rays_reach_the_triangle_they_hit(name, n, request, wide_lib):
    """Object-space rays against single BLASes placed with the identity transform."""
    from pt_amd import abi as A
    scene = request.getfixturevalue(name)
    orc = O.OracleScene(scene)
    b = max(orc.blases, key=lambda x: x.prim_count)
    ident = np.eye(4, dtype=np.float32)[:3, :]
    tl = O.Tlas(np.stack([A.make_instance(ident, 0, b.h)]).astype(A.INSTANCE), [b])
    rlo, rhi = b.root_box()
    rng = np.random.default_rng(11)
    c, ext = (rlo + rhi) / 2, float(np.max(rhi - rlo))
    o = (c + rng.uniform(-1.0, 1.0, size=(n, 3)) * ext).astype(np.float32)
    tgt = (c + rng.uniform(-0.5, 0.5, size=(n, 3)) * (rhi - rlo)).astype(np.float32)
    d = tgt - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    hits, t = tl.intersect(o, d)
    wide, _ = collapse(wide_lib, b.nodes())
    leaf_prims = b.leaf_prims()
    codes = np.zeros(b.prim_count + 8, np.uint32)
    n_hit = 0
    for i in range(n):
        if hits["instance"][i] == 0xFFFFFFFF:
            continue
        n_hit += 1
        oi, di = np.ascontiguousarray(o[i]), np.ascontiguousarray(d[i])
        m = wide_lib.pt_wide_candidates_host(wide.ctypes.data, 0, oi.ctypes.data, di.ctypes.data, 0.001, 10000.0, codes.ctypes.data, len(codes))
        assert m <= len(codes)
        reach = set()
        for code in codes[:m]:
            first, cnt = int(code) >> 3, (int(code) & 7) + 1
            reach.update(leaf_prims[first:first + cnt].tolist())
        assert int(hits["primitive"][i]) in reach
    assert n_hit > n // 20


def prim_boxes_in_index_order(blas):
    """Padded primitive boxes [n, 6] in primitive order, in binary32 like the builders compute them."""
    v = blas.vertices["pos"]
    idx = blas.indices[:3 * blas.prim_count].reshape(-1, 3)
    p = v[idx]
    d = np.float32(blas.delta())
    return np.ascontiguousarray(np.concatenate([(p.min(axis=1) - d), (p.max(axis=1) + d)], axis=1).astype(np.float32))


def sah_host(lib, boxes, max_leaf=4):
    n = len(boxes)
    lo = np.ascontiguousarray(boxes[:, :3].min(axis=0)); hi = np.ascontiguousarray(boxes[:, 3:].max(axis=0))
    nodes = np.zeros(n, dtype=abi.BVH_NODE)
    order = np.zeros(n, dtype=np.uint32)
    cnt = lib.pt_sah_build_host_c(boxes.ctypes.data, n, max_leaf, lo.ctypes.data, hi.ctypes.data, nodes.ctypes.data, order.ctypes.data)
    return nodes[:cnt].copy(), order


def check_binary_tree(nodes, order, boxes, max_leaf):
    """Every primitive in exactly one leaf, every node referenced once, child boxes = exact unions."""
    n = len(order)
    assert sorted(order.tolist()) == list(range(n))
    seen = np.zeros(n, np.int32)
    node_seen = np.zeros(len(nodes), np.int32)
    node_seen[0] = 1

    def visit(i):
        nd = nodes[i]
        lo_all, hi_all = None, None
        for c, lo, hi in ((int(nd["c0"]), nd["lo0"], nd["hi0"]), (int(nd["c1"]), nd["lo1"], nd["hi1"])):
            assert c != EMPTY
            if c >= 0:
                node_seen[c] += 1
                blo, bhi = visit(c)
            else:
                code = (~c) & 0xFFFFFFFF
                first, cnt = code >> 3, (code & 7) + 1
                assert cnt <= max_leaf
                seen[first:first + cnt] += 1
                b = boxes[order[first:first + cnt]]
                blo, bhi = b[:, :3].min(axis=0), b[:, 3:].max(axis=0)
            assert np.array_equal(blo, lo) and np.array_equal(bhi, hi)
            lo_all = blo if lo_all is None else np.minimum(lo_all, blo)
            hi_all = bhi if hi_all is None else np.maximum(hi_all, bhi)
        return lo_all, hi_all

    import sys
    sys.setrecursionlimit(100000)
    visit(0)
    assert np.all(seen == 1) and np.all(node_seen == 1)


def sah_cost(nodes):
    """Sum of child-box areas of inner children relative to the root box."""
    def area(lo, hi):
        d = np.maximum(hi - lo, 0).astype(np.float64)
        return d[..., 0] * d[..., 1] + d[..., 1] * d[..., 2] + d[..., 2] * d[..., 0]
    root = area(np.minimum(nodes["lo0"][0], nodes["lo1"][0]), np.maximum(nodes["hi0"][0], nodes["hi1"][0]))
    c = 1.0
    for ck, lo, hi in (("c0", "lo0", "hi0"), ("c1", "lo1", "hi1")):
        inner = nodes[ck] >= 0
        c += area(nodes[lo][inner], nodes[hi][inner]).sum() / root
    return c


@pytest.mark.parametrize("name", ["cornell", "cube", "soup", "dragon", "helmet"])
def test_sah_tree_is_a_partition_with_exact_boxes(name, request, wide_lib):
    scene = request.getfixturevalue(name)
    orc = O.OracleScene(scene)
    done = set()
    for b in orc.blases:
        key = (b.indices.ctypes.data, b.prim_count)
        if key in done or b.prim_count < 2:
            continue
        done.add(key)
        boxes = prim_boxes_in_index_order(b)
        nodes, order = sah_host(wide_lib, boxes)
        check_binary_tree(nodes, order, boxes, 4)
        # and the wide nodes derived from it are conservative
        wide, depth = collapse(wide_lib, nodes)
        lo, hi = boxes[order][:, :3], boxes[order][:, 3:]
        check_wide(wide, depth, lo, hi, b.prim_count)
        if b.prim_count > 1000:
            assert sah_cost(nodes) < 0.92 * sah_cost(b.nodes())      # what the second tree is for


def sweep_host(lib, boxes):
    n = len(boxes)
    nodes = np.zeros(n, dtype=abi.BVH_NODE)
    order = np.zeros(n, dtype=np.uint32)
    lib.pt_sweep_sah_build_host_c.restype = C.c_uint32
    lib.pt_sweep_sah_build_host_c.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    cnt = lib.pt_sweep_sah_build_host_c(boxes.ctypes.data, n, nodes.ctypes.data, order.ctypes.data)
    return nodes[:cnt].copy(), order


def _instance_boxes(orc):
    return np.ascontiguousarray(np.stack([np.concatenate(orc.tlas.instance_box(i)) for i in range(orc.tlas.n)]).astype(np.float32))


@pytest.mark.parametrize("name", ["cornell", "soup", "helmet", "stacked"])
def test_tlas_sweep_tree_is_a_partition_with_exact_boxes(name, request, wide_lib):
    """The full-sweep SAH tree a TLAS may be traversed through: one instance per leaf, every instance in exactly one
    leaf, child boxes the exact unions of the instance boxes below them, wide nodes conservative; and it never costs
    more than splitting every range in the middle of its original order."""
    scene = request.getfixturevalue(name)
    orc = O.OracleScene(scene)
    boxes = _instance_boxes(orc)
    if len(boxes) < 2:
        pytest.skip("one instance: no tree")
    nodes, order = sweep_host(wide_lib, boxes)
    assert len(nodes) == len(boxes) - 1
    check_binary_tree(nodes, order, boxes, 1)
    wide, depth = collapse(wide_lib, nodes)
    check_wide(wide, depth, boxes[order][:, :3], boxes[order][:, 3:], len(boxes))
    # the same boxes in a random order give a tree of the same cost class (the sweep looks at every partition along the axes)
    rng = np.random.default_rng(5)
    perm = rng.permutation(len(boxes))
    nodes2, order2 = sweep_host(wide_lib, np.ascontiguousarray(boxes[perm]))
    check_binary_tree(nodes2, order2, boxes[perm], 1)
    wide_lib.pt_tree_cost_host_c.restype = C.c_double
    wide_lib.pt_tree_cost_host_c.argtypes = [C.c_void_p, C.c_uint32]
    c1 = wide_lib.pt_tree_cost_host_c(nodes.ctypes.data, len(nodes)); c2 = wide_lib.pt_tree_cost_host_c(nodes2.ctypes.data, len(nodes2))
    assert abs(c1 - c2) <= 0.15 * max(c1, c2)
    assert abs(c1 - (sah_cost(nodes) - 1.0 + _leaf_child_area(nodes))) < 1e-3 * c1


def _leaf_child_area(nodes):
    """Areas of the leaf children relative to the root (sah_cost counts the inner children only)."""
    def area(lo, hi):
        d = np.maximum(hi - lo, 0).astype(np.float64)
        return d[..., 0] * d[..., 1] + d[..., 1] * d[..., 2] + d[..., 2] * d[..., 0]
    root = area(np.minimum(nodes["lo0"][0], nodes["lo1"][0]), np.maximum(nodes["hi0"][0], nodes["hi1"][0]))
    c = 0.0
    for ck, lo, hi in (("c0", "lo0", "hi0"), ("c1", "lo1", "hi1")):
        leaf = nodes[ck] < 0
        c += area(nodes[lo][leaf], nodes[hi][leaf]).sum() / root
    return c


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cornell", "helmet", "soup", "stacked"])
def test_gpu_tlas_traversal_tree_is_the_cheaper_one(name, request, wide_lib, monkeypatch):
    """pt_tlas_build traverses the cheaper of the Morton tree and the host's full-sweep SAH tree; forced either way,
    frames are the same bits."""
    from pt_amd import hip
    scene = request.getfixturevalue(name)
    orc = O.OracleScene(scene)
    boxes = _instance_boxes(orc)
    wide_lib.pt_tree_cost_host_c.restype = C.c_double
    wide_lib.pt_tree_cost_host_c.argtypes = [C.c_void_p, C.c_uint32]
    cost = lambda nd: wide_lib.pt_tree_cost_host_c(np.ascontiguousarray(nd).ctypes.data, len(nd))
    frames = {}
    for forced in (None, "lbvh", "sweep"):
        if forced:
            monkeypatch.setenv("PT_HIP_TLAS_TREE", forced)
        else:
            monkeypatch.delenv("PT_HIP_TLAS_TREE", raising=False)
        r = hip.Renderer(scene, 48, 32)
        try:
            morton = r.tlas.download()[0]
            nodes, order = r.tlas.download_traversal_tree()
            want_nodes, want_order = sweep_host(wide_lib, boxes)
            if forced == "lbvh":
                assert nodes.tobytes() == morton.tobytes()
            elif forced == "sweep":
                assert nodes.tobytes() == want_nodes.tobytes() and np.array_equal(order, want_order)
            else:
                cheaper_is_sweep = cost(want_nodes) < cost(morton)
                assert nodes.tobytes() == (want_nodes if cheaper_is_sweep else morton).tobytes()
            check_binary_tree(nodes, order, boxes, 1)
            r.render(spp=3, frame=1)
            frames[forced] = r.last_result().copy()
        finally:
            r.close()
    assert np.array_equal(frames[None].view(np.uint32), frames["lbvh"].view(np.uint32))
    assert np.array_equal(frames[None].view(np.uint32), frames["sweep"].view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cornell", "soup", "dragon", "stacked", "helmet"])
def test_gpu_sah_builder_matches_host_twin(name, request, wide_lib):
    """Same decisions, level order and numbering on both sides: nodes and leaf order agree byte for byte."""
    from pt_amd import hip
    import oracle as O2
    scene = request.getfixturevalue(name)
    orc = O2.OracleScene(scene)
    r = hip.Renderer(scene, 32, 32)
    try:
        for b, ob in zip(r.blases, orc.blases):
            if ob.prim_count < 2:
                continue
            nodes, order = b.download_traversal_tree()
            want_nodes, want_order = sah_host(wide_lib, prim_boxes_in_index_order(ob))
            assert np.array_equal(order, want_order)
            assert nodes.tobytes() == want_nodes.tobytes()
    finally:
        r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cornell", "soup", "dragon"])
def test_gpu_builder_matches_host_collapse(name, request, wide_lib):
    """The GPU derives the wide nodes level by level with a scan; the host helper walks the same
    binary nodes breadth first.  Same per-node routine, same numbering: the bytes must agree."""
    from pt_amd import hip
    scene = request.getfixturevalue(name)
    r = hip.Renderer(scene, 32, 32)
    try:
        for b in r.blases:
            nodes, _ = b.download_traversal_tree()
            got, depth = b.download_wide()
            want, want_depth = collapse(wide_lib, nodes)
            assert depth == want_depth and got.tobytes() == want.tobytes()
        nodes = r.tlas.download_traversal_tree()[0]
        got, depth = r.tlas.download_wide()
        want, want_depth = collapse(wide_lib, nodes)
        assert depth == want_depth and got.tobytes() == want.tobytes()
    finally:
        r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["dragon", "helmet"])
def test_gpu_traversal_tree_quality_and_reach(name, request, wide_lib):
    """The trees the GPU built, downloaded and judged without any of the builders' code: the binned-SAH tree
    costs less (expected box tests) than the Morton tree over the same primitives, the wide nodes derived
    from it partition the triangles conservatively, and a plain host walk of those wide nodes reaches
    every triangle that brute force (no tree at all) says a ray hits."""
    from pt_amd import abi as A
    from pt_amd import hip
    scene = request.getfixturevalue(name)
    orc = O.OracleScene(scene)
    r = hip.Renderer(scene, 32, 32)
    try:
        big = [k for k, ob in enumerate(orc.blases) if ob.prim_count > 5000]
        assert big
        for k in big:
            b, ob = r.blases[k], orc.blases[k]
            sah_nodes, order = b.download_traversal_tree()
            morton_nodes, _, _ = b.download()
            c_sah, c_morton = sah_cost(sah_nodes), sah_cost(morton_nodes)
            assert c_sah < 0.92 * c_morton, (c_sah, c_morton)
            boxes = prim_boxes_in_index_order(ob)
            check_binary_tree(sah_nodes, order, boxes, 4)
            wide, depth = b.download_wide()
            check_wide(wide, depth, boxes[order][:, :3], boxes[order][:, 3:], ob.prim_count)
        # rays against the largest BLAS under the identity transform
        k = max(big, key=lambda i: orc.blases[i].prim_count)
        b, ob = r.blases[k], orc.blases[k]
        wide, _ = b.download_wide()
        _, order = b.download_traversal_tree()
        ident = np.eye(4, dtype=np.float32)[:3, :]
        tl = O.Tlas(np.stack([A.make_instance(ident, 0, ob.h)]).astype(A.INSTANCE), [ob])
        rlo, rhi = ob.root_box()
        rng = np.random.default_rng(23)
        n = 120
        c, ext = (rlo + rhi) / 2, float(np.max(rhi - rlo))
        o = (c + rng.uniform(-1.0, 1.0, size=(n, 3)) * ext).astype(np.float32)
        tgt = (c + rng.uniform(-0.4, 0.4, size=(n, 3)) * (rhi - rlo)).astype(np.float32)
        d = tgt - o
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        hits, _ = tl.intersect(o, d, use_bvh=False)
        codes = np.zeros(ob.prim_count + 8, np.uint32)
        n_hit = 0
        for i in range(n):
            if hits["instance"][i] == 0xFFFFFFFF:
                continue
            n_hit += 1
            oi, di = np.ascontiguousarray(o[i]), np.ascontiguousarray(d[i])
            m = wide_lib.pt_wide_candidates_host(wide.ctypes.data, 0, oi.ctypes.data, di.ctypes.data, 0.001, 10000.0, codes.ctypes.data, len(codes))
            reach = set()
            for code in codes[:min(m, len(codes))]:
                first, cnt = int(code) >> 3, (int(code) & 7) + 1
                reach.update(order[first:first + cnt].tolist())
            assert int(hits["primitive"][i]) in reach
        assert n_hit > n // 10
    finally:
        r.close()


@pytest.mark.gpu
def test_builders_give_the_same_trees_without_the_polled_mailbox(dragon, monkeypatch):
    """The per-level counts of the SAH and wide-node builders reach the host through a mapped word it polls;
    where such memory cannot be had they go through a device word, a copy and a stream wait.  Both ways must
    build the same bytes."""
    from pt_amd import hip

    def trees():
        r = hip.Renderer(dragon, 16, 16)
        try:
            out = []
            for b in r.blases:
                nodes, order = b.download_traversal_tree()
                wide, depth = b.download_wide()
                out.append((nodes.tobytes(), order.tobytes(), np.asarray(wide).tobytes(), depth))
            out.append(np.asarray(r.tlas.download_wide()[0]).tobytes() if hasattr(r.tlas, "download_wide") else b"")
            return out
        finally:
            r.close()

    polled = trees()
    monkeypatch.setenv("PT_HIP_NO_MAILBOX", "1")
    assert trees() == polled
