#!/usr/bin/env python3
"""Fill statistics of the 4-wide traversal nodes of the bench scene (run on the GPU box)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
from pt_amd import hip, procedural

scene = procedural.dragon_standin()
r = hip.Renderer(scene, 64, 64, max_leaf=int(os.environ.get("MAX_LEAF", "0")))
for b in r.blases:
    wide, depth = b.download_wide()
    child = wide["child"]
    empty = child == np.int32(-2**31)
    leaf = (child < 0) & ~empty
    inner = child >= 0
    cnt = ((~child[leaf]).astype(np.uint32) & 7) + 1
    per_node = (~empty).sum(axis=1)
    print(f"nodes {len(wide)} depth {depth} children/node {per_node.mean():.2f} hist {np.bincount(per_node, minlength=5).tolist()} "
          f"leaves {leaf.sum()} inner refs {inner.sum()} tris {cnt.sum()} tris/leaf {cnt.mean():.2f} hist {np.bincount(cnt, minlength=9).tolist()}")
r.close()
