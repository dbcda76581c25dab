"""Writes a Scene back out as glTF 2.0 (one embedded base64 buffer, PNG images next to it).

Used to exercise the importers on machines where the reference's assets are not present: the file
it writes imports to exactly the arrays it was written from (the importers' double V flip,
1 - (1 - v), leaves values that already went through it unchanged)."""
import base64
import json
import os

import numpy as np

from . import abi


def save_gltf(scene, path):
    assert len(scene.models) == 1
    m = scene.models[0]
    d = os.path.dirname(os.path.abspath(path))
    blob = bytearray()
    views, accessors = [], []

    def add(arr, ctype, atype):
        off = len(blob)
        blob.extend(np.ascontiguousarray(arr).tobytes())
        while len(blob) % 4:
            blob.append(0)
        views.append({"buffer": 0, "byteOffset": off, "byteLength": arr.nbytes})
        accessors.append({"bufferView": len(views) - 1, "componentType": ctype, "count": int(arr.shape[0]), "type": atype})
        return len(accessors) - 1

    images = []
    from PIL import Image
    for i, img in enumerate(scene.tables.images):
        if i == 0:
            continue
        name = f"{os.path.splitext(os.path.basename(path))[0]}_image{i}.png"
        Image.fromarray(img, "RGBA").save(os.path.join(d, name))
        images.append({"uri": name})
    mats = []
    table = scene.tables.material_array()
    for h in m.materials:
        t = table[h]
        g = {"pbrMetallicRoughness": {"baseColorFactor": [float(v) for v in t["albedoFactor"]],
                                      "metallicFactor": float(t["metallicFactor"]), "roughnessFactor": float(t["roughnessFactor"])},
             "emissiveFactor": [float(v) for v in t["emissiveFactor"]]}
        if t["useAlbedoMap"]:
            g["pbrMetallicRoughness"]["baseColorTexture"] = {"index": int(t["albedoMapIndex"]) - 1}
        if t["useMRMap"]:
            g["pbrMetallicRoughness"]["metallicRoughnessTexture"] = {"index": int(t["mrMapIndex"]) - 1}
        mats.append(g)
    prims = []
    for mesh in m.meshes:
        idx = m.indices[mesh.first_index:mesh.first_index + mesh.index_count].astype(np.uint32) - np.uint32(mesh.first_vertex)
        nv = int(idx.max()) + 1
        v = m.vertices[mesh.first_vertex:mesh.first_vertex + nv]
        uv = v["uv"].copy()   # stored as is: the importer's 1 - (1 - v) is idempotent on imported values
        prims.append({"attributes": {"POSITION": add(v["pos"], 5126, "VEC3"), "NORMAL": add(v["normal"], 5126, "VEC3"),
                                     "TEXCOORD_0": add(uv, 5126, "VEC2")},
                      "indices": add(idx, 5125, "SCALAR"), "material": m.materials.index(mesh.material), "mode": 4})
    gl_meshes, nodes = [], []
    for n in m.nodes:
        g = {"name": n.name, "matrix": [float(x) for x in np.asarray(n.local_matrix, dtype=np.float32).T.reshape(16)]}
        if n.meshes:
            gl_meshes.append({"primitives": [prims[k] for k in n.meshes]})
            g["mesh"] = len(gl_meshes) - 1
        nodes.append(g)
    for i, n in enumerate(m.nodes):
        if n.parent is not None:
            nodes[n.parent].setdefault("children", []).append(i)
    roots = [i for i, n in enumerate(m.nodes) if n.parent is None]
    j = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": roots}], "nodes": nodes, "meshes": gl_meshes,
         "materials": mats, "accessors": accessors, "bufferViews": views,
         "buffers": [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(bytes(blob)).decode()}]}
    if images:
        j["images"] = images
        j["textures"] = [{"source": k} for k in range(len(images))]
    with open(path, "w") as f:
        json.dump(j, f)
