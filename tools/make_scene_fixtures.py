#!/usr/bin/env python3
"""Converts the reference's small glTF assets into .npz scene fixtures under tests/golden/.

Run in the build container, where the reference checkout is mounted:
    python tools/make_scene_fixtures.py /root/reference/assets
The fixtures hold data only (vertex/index arrays, node matrices, material table, RGBA8 images)
as imported by pt_amd.gltf with the reference's Assimp conventions.
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "path-tracer_amd"))
from pt_amd import gltf  # noqa: E402
from pt_amd.scene import Scene  # noqa: E402


def main(assets):
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
    os.makedirs(out, exist_ok=True)
    for name, rel in (("cornell", "cornell/CornellBox-Original.gltf"), ("cube", "cube/Cube.gltf")):
        s = Scene()
        gltf.load_from_file(s, os.path.join(assets, rel))
        s.save_npz(os.path.join(out, f"scene_{name}.npz"))
        print(name, "instances", len(s.instances()), "triangles", s.triangle_count(), "images", len(s.tables.images))
    # The helmet with its textures reduced: base-colour maps box-filtered to 256 x 256, the maps no
    # shader samples (normal, occlusion/roughness/metallic) replaced by 2 x 2 zeros.  Geometry, nodes,
    # materials and texture indices are the import's; the fixture exercises textured shading on
    # 94,722 triangles in 5 instances, not the reference's texels.
    import numpy as np
    s = Scene()
    gltf.load_from_file(s, os.path.join(assets, "helmet/FlightHelmet.gltf"))
    mats = s.tables.material_array()
    used = {int(m["albedoMapIndex"]) for m in mats if m["useAlbedoMap"]}
    for k, im in enumerate(s.tables.images):
        if k == 0:
            continue
        if k in used:
            f = im.shape[0] // 256
            s.tables.images[k] = np.ascontiguousarray(im.reshape(256, f, 256, f, 4).astype(np.uint32).mean(axis=(1, 3)).round().astype(np.uint8))
        else:
            s.tables.images[k] = np.zeros((2, 2, 4), np.uint8)
    s.save_npz(os.path.join(out, "scene_helmet.npz"))
    print("helmet", "instances", len(s.instances()), "triangles", s.triangle_count(), "images", len(s.tables.images))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/assets")
