"""The 8-wide layout of pt_wide8.h (evaluated on the host only, see sim_layouts.py): partition and containment.
Not collected with tests/: run as  python -m pytest experimental/check_wide8.py  after  make -C experimental."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import *   # noqa: E402,F401,F403  (scene fixtures, import paths)
from test_wide_nodes import O, prim_boxes_in_index_order, sah_host   # noqa: E402


@pytest.fixture(scope="module")
def wide_lib():
    lib = C.CDLL(os.path.join(ROOT, "experimental", "_build", "libpt_layout_sim.so"))
    lib.pt_sah_build_host_c.restype = C.c_uint32
    lib.pt_sah_build_host_c.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return lib


WIDE8 = np.dtype([("origin", "<f4", 3), ("exp_imask", "<u4"), ("child_base", "<u4"), ("leaf_base", "<u4"), ("lmask", "<u4"), ("pad", "<u4"),
                  ("qlo", "u1", (3, 8)), ("qhi", "u1", (3, 8))])


@pytest.mark.parametrize("name", ["cube", "soup", "dragon"])
def test_wide8_collapse_is_a_conservative_partition(name, request, wide_lib):
    """Implicit child references: inner children consecutive from child_base, leaf items consecutive from
    leaf_base, both in slot order.  Every node but the root and every primitive is referenced exactly once, and
    every child's dequantised box contains what lies below it."""
    assert WIDE8.itemsize == 80
    scene = request.getfixturevalue(name)
    orc = O.OracleScene(scene)
    b = max(orc.blases, key=lambda x: x.prim_count)
    boxes = prim_boxes_in_index_order(b)
    nodes, order = sah_host(wide_lib, boxes, max_leaf=1)
    out = np.zeros(len(nodes) + 1, dtype=WIDE8)
    order8 = np.zeros(b.prim_count, np.uint32)
    depth = C.c_uint32(0)
    wide_lib.pt_wide8_collapse_host.restype = C.c_uint32
    n = wide_lib.pt_wide8_collapse_host(C.c_void_p(nodes.ctypes.data), C.c_uint32(len(nodes)), C.c_uint32(0), C.c_void_p(out.ctypes.data),
                                        C.c_uint32(len(out)), C.c_void_p(order8.ctypes.data), C.byref(depth))
    assert n > 0
    w8 = out[:n]
    prim_of_item = order[order8]
    assert sorted(prim_of_item.tolist()) == list(range(b.prim_count))
    node_refs = np.zeros(n, np.int32)
    item_refs = np.zeros(b.prim_count, np.int32)

    def child_box(w, s):
        lo, hi = np.zeros(3, np.float32), np.zeros(3, np.float32)
        for k in range(3):
            scale = np.float64(2.0) ** (int((int(w["exp_imask"]) >> (8 * k)) & 255) - 127)
            lo[k] = np.float32(np.float64(w["qlo"][k][s]) * scale + np.float64(w["origin"][k]))
            hi[k] = np.float32(np.float64(w["qhi"][k][s]) * scale + np.float64(w["origin"][k]))
        return lo, hi

    def visit(i, d):
        w = w8[i]
        imask, lmask = int(w["exp_imask"]) >> 24, int(w["lmask"])
        assert imask & lmask == 0 and (imask | lmask) != 0
        lo_all, hi_all = np.full(3, np.inf, np.float32), np.full(3, -np.inf, np.float32)
        ri = rl = 0
        for s in range(8):
            if (imask >> s) & 1:
                c = int(w["child_base"]) + ri; ri += 1
                node_refs[c] += 1
                blo, bhi = visit(c, d + 1)
            elif (lmask >> s) & 1:
                item = int(w["leaf_base"]) + rl; rl += 1
                item_refs[item] += 1
                bx = boxes[prim_of_item[item]]
                blo, bhi = bx[:3], bx[3:]
            else:
                assert int(w["qlo"][0][s]) == 255 and int(w["qhi"][0][s]) == 0      # an empty slot keeps the inverted box
                continue
            lo, hi = child_box(w, s)
            assert np.all(lo <= blo) and np.all(hi >= bhi)
            lo_all, hi_all = np.minimum(lo_all, blo), np.maximum(hi_all, bhi)
        assert d <= depth.value
        return lo_all, hi_all

    import sys
    sys.setrecursionlimit(100000)
    visit(0, 1)
    assert np.all(node_refs[1:] == 1) and node_refs[0] == 0 and np.all(item_refs == 1)
