"""glTF 2.0 import with the conventions the reference gets from Assimp.

Mirrors ModelLoader::LoadFromFile / ProcessModel (source/model_loader.cpp:273-362), which calls
Assimp 6.0.1 with aiProcess_Triangulate | aiProcess_FlipUVs | aiProcess_GenNormals (:277):
  - one mesh per glTF primitive, in mesh order then primitive order; a node lists the meshes of
    all primitives of its glTF mesh;
  - vertices appended per mesh, indices rebased by mesh.firstVertex (:131-150);
  - nodes in depth-first pre-order with parent links (:193-251); a scene with several root
    nodes gets a synthetic identity root;
  - node TRS composed in binary32 as T * R * S with the quaternion (x, y, z, w);
  - Assimp's glTF importer stores v' = 1 - v and aiProcess_FlipUVs flips again, so the texture
    coordinate that reaches the shader is 1 - (1 - v) evaluated in binary32;
  - per material (:57-126): textures are loaded in the order albedo, metallic-roughness, normal,
    occlusion, emissive; every successful load appends a new image (there is no working cache,
    :43-55); a file that cannot be read yields the null handle 0xFFFF.
Assimp itself is not available offline; behaviours above that the reference never exercises in
its default scene are "parity unpinned".
"""
import base64
import json
import os

import numpy as np

from . import abi
from .scene import Mesh, Model, Node, Scene, f32

_COMPONENT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def _load_image_rgba8(path):
    """stbi_load(path, ..., 4) equivalent (source/model_loader.cpp:25): RGBA8, top row first."""
    try:
        from PIL import Image
        with Image.open(path) as im:
            return np.ascontiguousarray(np.asarray(im.convert("RGBA"), dtype=np.uint8))
    except Exception:
        return None


class _Gltf:
    def __init__(self, path):
        self.dir = os.path.dirname(path)
        with open(path, "r") as f:
            self.j = json.load(f)
        self.buffers = []
        for b in self.j.get("buffers", []):
            uri = b["uri"]
            if uri.startswith("data:"):
                self.buffers.append(base64.b64decode(uri.split(",", 1)[1]))
            else:
                with open(os.path.join(self.dir, uri), "rb") as f:
                    self.buffers.append(f.read())

    def accessor(self, idx):
        a = self.j["accessors"][idx]
        bv = self.j["bufferViews"][a["bufferView"]]
        dt = np.dtype(_COMPONENT[a["componentType"]]).newbyteorder("<")
        nc = _NCOMP[a["type"]]
        off = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
        stride = bv.get("byteStride", 0) or dt.itemsize * nc
        buf = self.buffers[bv["buffer"]]
        out = np.empty((a["count"], nc), dtype=dt)
        for c in range(nc):
            out[:, c] = np.ndarray((a["count"],), dtype=dt, buffer=buf, offset=off + c * dt.itemsize, strides=(stride,))
        return out


def _trs_matrix(node):
    """aiNode::mTransformation as Assimp's glTF2 importer builds it (binary32)."""
    if "matrix" in node:
        return np.array(node["matrix"], dtype=f32).reshape(4, 4).T.copy()  # glTF stores column-major
    m = np.eye(4, dtype=f32)
    if "translation" in node:
        t = np.eye(4, dtype=f32)
        t[0:3, 3] = np.array(node["translation"], dtype=f32)
        m = _mul(m, t)
    if "rotation" in node:
        x, y, z, w = [f32(v) for v in node["rotation"]]
        r = np.eye(4, dtype=f32)
        one, two = f32(1.0), f32(2.0)
        r[0, 0] = one - two * (y * y + z * z); r[0, 1] = two * (x * y - z * w); r[0, 2] = two * (x * z + y * w)
        r[1, 0] = two * (x * y + z * w); r[1, 1] = one - two * (x * x + z * z); r[1, 2] = two * (y * z - x * w)
        r[2, 0] = two * (x * z - y * w); r[2, 1] = two * (y * z + x * w); r[2, 2] = one - two * (x * x + y * y)
        m = _mul(m, r)
    if "scale" in node:
        s = np.eye(4, dtype=f32)
        s[0, 0], s[1, 1], s[2, 2] = [f32(v) for v in node["scale"]]
        m = _mul(m, s)
    return m


def _mul(a, b):
    from .scene import mat4_mul
    return mat4_mul(a, b)


def load_from_file(scene: Scene, path: str) -> Model:
    """ModelLoader::LoadFromFile: appends images/materials to scene.tables and the model to scene.models."""
    g = _Gltf(path)
    j = g.j
    tables = scene.tables
    model = Model(vertices=np.zeros((0,), dtype=abi.VERTEX), indices=np.zeros((0,), dtype=np.uint32),
                  name=os.path.basename(path))

    def texture_path(tex_info):
        if tex_info is None:
            return None
        tex = j["textures"][tex_info["index"]]
        return j["images"][tex["source"]].get("uri")

    def load_texture(uri):
        if uri is None:
            return abi.NULL_RESOURCE_INDEX_VALUE
        img = _load_image_rgba8(os.path.join(g.dir, uri))
        if img is None:
            handle = abi.NULL_RESOURCE_INDEX_VALUE
        else:
            handle = tables.create_image(img)
        model.textures.append(handle)
        return handle

    # ---- materials (ProcessMaterial, source/model_loader.cpp:57-126)
    for gm in j.get("materials", []):
        m = abi.default_material()
        pbr = gm.get("pbrMetallicRoughness", {})
        albedo = load_texture(texture_path(pbr.get("baseColorTexture")))
        mr = load_texture(texture_path(pbr.get("metallicRoughnessTexture")))
        normal = load_texture(texture_path(gm.get("normalTexture")))
        # Assimp files glTF occlusion textures under aiTextureType_LIGHTMAP, so the reference's
        # aiTextureType_AMBIENT_OCCLUSION query (:80) finds nothing.
        occlusion = abi.NULL_RESOURCE_INDEX_VALUE
        emissive = load_texture(texture_path(gm.get("emissiveTexture")))
        for key, use, handle in (("albedoMapIndex", "useAlbedoMap", albedo), ("mrMapIndex", "useMRMap", mr),
                                 ("normalMapIndex", "useNormalMap", normal), ("occlusionMapIndex", "useOcclusionMap", occlusion),
                                 ("emissiveMapIndex", "useEmissiveMap", emissive)):
            m[key] = handle
            m[use] = int(handle != abi.NULL_RESOURCE_INDEX_VALUE)
        m["albedoFactor"] = np.array(pbr.get("baseColorFactor", [1, 1, 1, 1]), dtype=f32)
        m["metallicFactor"] = f32(pbr.get("metallicFactor", 1.0))
        m["roughnessFactor"] = f32(pbr.get("roughnessFactor", 1.0))
        if "normalTexture" in gm:
            m["normalScale"] = f32(gm["normalTexture"].get("scale", 1.0))
        m["emissiveFactor"] = np.array(gm.get("emissiveFactor", [0, 0, 0]), dtype=f32)
        if "occlusionTexture" in gm:
            m["occlusionStrength"] = f32(gm["occlusionTexture"].get("strength", 1.0))
        model.materials.append(tables.create_material(m))

    # ---- meshes (ProcessMesh, :128-191): one per primitive
    verts, inds = [], []
    nverts = nindices = 0
    mesh_of_prim = []  # glTF mesh index -> list of our mesh indices
    for gmesh in j.get("meshes", []):
        ids = []
        for prim in gmesh["primitives"]:
            if prim.get("mode", 4) != 4:
                raise ValueError("only triangle lists are imported")
            attr = prim["attributes"]
            pos = g.accessor(attr["POSITION"]).astype(f32)
            n = pos.shape[0]
            v = np.zeros((n,), dtype=abi.VERTEX)
            v["pos"] = pos
            if "NORMAL" in attr:
                v["normal"] = g.accessor(attr["NORMAL"]).astype(f32)
            if "TEXCOORD_0" in attr:
                uv = g.accessor(attr["TEXCOORD_0"]).astype(f32)
                v["uv"][:, 0] = uv[:, 0]
                v["uv"][:, 1] = f32(1.0) - (f32(1.0) - uv[:, 1])
            if "indices" in prim:
                idx = g.accessor(prim["indices"]).astype(np.uint32).reshape(-1)
            else:
                idx = np.arange(n, dtype=np.uint32)
            if "NORMAL" not in attr:
                v["normal"] = _gen_face_normals(pos, idx)
            mat_index = prim.get("material")
            handle = model.materials[mat_index] if mat_index is not None and mat_index < len(model.materials) else abi.NULL_RESOURCE_INDEX_VALUE
            ids.append(len(model.meshes))
            model.meshes.append(Mesh(index_count=int(idx.size), first_index=nindices, first_vertex=nverts, material=handle))
            inds.append(idx + np.uint32(nverts))
            verts.append(v)
            nverts += n
            nindices += int(idx.size)
        mesh_of_prim.append(ids)
    model.vertices = np.concatenate(verts) if verts else model.vertices
    model.indices = np.concatenate(inds).astype(np.uint32) if inds else model.indices

    # ---- nodes (ProcessNodes, :193-251)
    gnodes = j.get("nodes", [])
    roots = j["scenes"][j.get("scene", 0)]["nodes"]

    def visit(gi, parent):
        gn = gnodes[gi]
        me = len(model.nodes)
        model.nodes.append(Node(gn.get("name", f"node{gi}"), parent, _trs_matrix(gn),
                                list(mesh_of_prim[gn["mesh"]]) if "mesh" in gn else []))
        for c in gn.get("children", []):
            visit(c, me)

    if len(roots) == 1:
        visit(roots[0], None)
    else:
        model.nodes.append(Node("ROOT", None, np.eye(4, dtype=f32), []))
        for r in roots:
            visit(r, 0)
    scene.add_model(model)
    return model


def _gen_face_normals(pos, idx):
    """aiProcess_GenNormals stand-in: flat face normals, last writer wins per vertex."""
    nrm = np.zeros_like(pos)
    tri = idx.reshape(-1, 3)
    e1 = pos[tri[:, 1]] - pos[tri[:, 0]]
    e2 = pos[tri[:, 2]] - pos[tri[:, 0]]
    fn = np.cross(e1, e2)
    ln = np.linalg.norm(fn, axis=1, keepdims=True)
    fn = fn / np.where(ln > 0, ln, 1)
    for k in range(3):
        nrm[tri[:, k]] = fn
    return nrm.astype(f32)
