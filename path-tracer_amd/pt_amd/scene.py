"""Host-side scene containers: the reference's Model / Mesh / Node plus the bindless tables.

Mirrors include/model_loader.hpp:16-52 (Node, Mesh, Model) and the append-only tables of
include/resources/bindless_resources.hpp:43-93.  Everything is plain numpy; device upload is
done by whoever consumes the scene (the HIP binding or the CPU oracle used by the tests).
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import abi

f32 = np.float32


def mat4_mul(a, b):
    """glm-style mat4 product in binary32 (math convention: (a @ b)[r][c] = sum_k a[r][k] b[k][c],
    accumulated k = 0..3 in order)."""
    a = np.asarray(a, dtype=f32)
    b = np.asarray(b, dtype=f32)
    out = np.zeros((4, 4), dtype=f32)
    for r in range(4):
        for c in range(4):
            s = f32(a[r, 0] * b[0, c])
            for k in range(1, 4):
                s = f32(s + f32(a[r, k] * b[k, c]))
            out[r, c] = s
    return out


@dataclass
class Node:  # include/model_loader.hpp:16-24
    name: str
    parent: Optional[int]
    local_matrix: np.ndarray  # 4x4 float32, math (row, column) convention
    meshes: List[int] = field(default_factory=list)


@dataclass
class Mesh:  # include/model_loader.hpp:26-32
    index_count: int
    first_index: int
    first_vertex: int
    material: int  # handle into the global material table, 0xFFFF = null


@dataclass
class Model:  # include/model_loader.hpp:34-52
    vertices: np.ndarray  # (N,) abi.VERTEX
    indices: np.ndarray   # (M,) uint32, already rebased to the model-wide vertex array
    nodes: List[Node] = field(default_factory=list)
    meshes: List[Mesh] = field(default_factory=list)
    textures: List[int] = field(default_factory=list)
    materials: List[int] = field(default_factory=list)
    name: str = ""

    def world_matrix(self, node_index):
        """Node::GetWorldMatrix, source/model_loader.cpp:253-265."""
        n = self.nodes[node_index]
        m = n.local_matrix.astype(f32)
        p = n.parent
        while p is not None:
            m = mat4_mul(self.nodes[p].local_matrix, m)
            p = self.nodes[p].parent
        return m


class BindlessTables:
    """Append-only image / material tables (include/resources/bindless_resources.hpp,
    source/resources/bindless_resources.cpp:38-62): image 0 is a 2x2 all-zero fallback."""

    def __init__(self):
        self.images: List[np.ndarray] = [np.zeros((2, 2, 4), dtype=np.uint8)]
        self.materials: List[np.ndarray] = []

    def create_image(self, rgba8):
        assert rgba8.dtype == np.uint8 and rgba8.ndim == 3 and rgba8.shape[2] == 4
        self.images.append(np.ascontiguousarray(rgba8))
        return len(self.images) - 1

    def create_material(self, m):
        self.materials.append(np.array(m, dtype=abi.MATERIAL))
        return len(self.materials) - 1

    def material_array(self):
        if not self.materials:
            return np.zeros((0,), dtype=abi.MATERIAL)
        return np.stack(self.materials).astype(abi.MATERIAL)


@dataclass
class InstanceRef:
    """One (model, node, mesh) tuple in Renderer::InitializeBLAS order (source/renderer.cpp:457-470)."""
    model: int
    node: int
    mesh: int
    world: np.ndarray  # 4x4 float32


class Scene:
    """A list of models plus the shared tables; instance i = i-th (model, node-DFS, mesh-in-node)."""

    def __init__(self):
        self.tables = BindlessTables()
        self.models: List[Model] = []

    def add_model(self, model):
        self.models.append(model)
        return len(self.models) - 1

    def instances(self) -> List[InstanceRef]:
        out = []
        for mi, model in enumerate(self.models):
            for ni, node in enumerate(model.nodes):
                for mesh in node.meshes:
                    out.append(InstanceRef(mi, ni, mesh, model.world_matrix(ni)))
        return out

    def merge(self, other, root_matrix=None, name="merged"):
        """Appends the models of `other` (a deep copy: `other` is left as it was) with its images and
        materials renumbered into this scene's tables, each model's node tree hung under a new root
        node with `root_matrix`.  Instance order stays the reference's: models in the order they were
        added, nodes depth first (source/renderer.cpp:457-470)."""
        import copy
        img_map = {0: 0, 0xFFFF: 0xFFFF}
        for i, img in enumerate(other.tables.images):
            if i > 0:
                img_map[i] = self.tables.create_image(img.copy())
        mat_map = {0xFFFF: 0xFFFF}
        for i, m in enumerate(other.tables.materials):
            m2 = np.array(m, dtype=abi.MATERIAL)
            for f in ("albedoMapIndex", "mrMapIndex", "normalMapIndex", "occlusionMapIndex", "emissiveMapIndex"):
                m2[f] = img_map.get(int(m[f]), 0xFFFF)
            mat_map[i] = self.tables.create_material(m2)
        root = np.eye(4, dtype=f32) if root_matrix is None else np.asarray(root_matrix, dtype=f32)
        for mi, model in enumerate(other.models):
            m = copy.deepcopy(model)
            for mesh in m.meshes:
                mesh.material = mat_map.get(mesh.material, 0xFFFF)
            m.materials = [mat_map.get(x, 0xFFFF) for x in m.materials]
            m.textures = [img_map.get(x, 0xFFFF) for x in m.textures]
            m.nodes = [Node(f"{name}_root{mi}", None, root.copy(), [])] + \
                      [Node(n.name, 0 if n.parent is None else n.parent + 1, n.local_matrix, list(n.meshes)) for n in m.nodes]
            self.add_model(m)
        return self

    def triangle_count(self):
        return sum(m.meshes[i.mesh].index_count // 3 for i in self.instances() for m in [self.models[i.model]])

    # ---- fixtures ------------------------------------------------------------------------
    def save_npz(self, path):
        d = {"n_models": np.int32(len(self.models)), "materials": self.tables.material_array(),
             "n_images": np.int32(len(self.tables.images))}
        for i, img in enumerate(self.tables.images):
            d[f"image{i}"] = img
        for mi, m in enumerate(self.models):
            d[f"m{mi}_vertices"] = m.vertices
            d[f"m{mi}_indices"] = m.indices
            d[f"m{mi}_meshes"] = np.array([[x.index_count, x.first_index, x.first_vertex, x.material] for x in m.meshes], dtype=np.uint32).reshape(-1, 4)
            d[f"m{mi}_node_parent"] = np.array([-1 if n.parent is None else n.parent for n in m.nodes], dtype=np.int32)
            d[f"m{mi}_node_local"] = np.stack([n.local_matrix for n in m.nodes]).astype(f32)
            d[f"m{mi}_node_names"] = np.array([n.name for n in m.nodes], dtype="U64")
            flat = [(ni, me) for ni, n in enumerate(m.nodes) for me in n.meshes]
            d[f"m{mi}_node_meshes"] = np.array(flat, dtype=np.int32).reshape(-1, 2)
            d[f"m{mi}_materials"] = np.array(m.materials, dtype=np.uint32)
            d[f"m{mi}_textures"] = np.array(m.textures, dtype=np.uint32)
        np.savez_compressed(path, **d)

    def save_ptscene(self, path):
        """Little-endian container read by the C++ ModelLoader (host/model_loader.cpp: LoadPtScene):
        "PTSC1", images (index 0 = fallback), materials, one model."""
        import struct
        assert len(self.models) == 1, "a .ptscene holds exactly one model"
        m = self.models[0]
        with open(path, "wb") as f:
            f.write(b"PTSC1\0\0\0")
            f.write(struct.pack("<I", len(self.tables.images)))
            for img in self.tables.images:
                f.write(struct.pack("<II", img.shape[1], img.shape[0]))
                f.write(np.ascontiguousarray(img).tobytes())
            mats = self.tables.material_array()
            f.write(struct.pack("<I", len(mats)))
            f.write(mats.tobytes())
            f.write(struct.pack("<I", 1))
            f.write(struct.pack("<I", len(m.vertices)))
            f.write(m.vertices.tobytes())
            f.write(struct.pack("<I", len(m.indices)))
            f.write(m.indices.astype("<u4").tobytes())
            f.write(struct.pack("<I", len(m.meshes)))
            for x in m.meshes:
                f.write(struct.pack("<IIII", x.index_count, x.first_index, x.first_vertex, x.material))
            f.write(struct.pack("<I", len(m.nodes)))
            for n in m.nodes:
                f.write(struct.pack("<i", -1 if n.parent is None else n.parent))
                f.write(np.ascontiguousarray(n.local_matrix, dtype="<f4").tobytes())
                f.write(struct.pack("<I", len(n.meshes)))
                f.write(np.array(n.meshes, dtype="<u4").tobytes())
                name = n.name.encode()
                f.write(struct.pack("<I", len(name)))
                f.write(name)

    @staticmethod
    def load_npz(path):
        z = np.load(path, allow_pickle=False)
        s = Scene()
        s.tables.images = [z[f"image{i}"] for i in range(int(z["n_images"]))]
        s.tables.materials = [m for m in z["materials"]]
        for mi in range(int(z["n_models"])):
            model = Model(vertices=z[f"m{mi}_vertices"], indices=z[f"m{mi}_indices"])
            model.meshes = [Mesh(int(a), int(b), int(c), int(d)) for a, b, c, d in z[f"m{mi}_meshes"]]
            parents = z[f"m{mi}_node_parent"]
            locals_ = z[f"m{mi}_node_local"]
            names = z[f"m{mi}_node_names"]
            model.nodes = [Node(str(names[i]), None if parents[i] < 0 else int(parents[i]), locals_[i].astype(f32)) for i in range(len(parents))]
            for ni, me in z[f"m{mi}_node_meshes"]:
                model.nodes[int(ni)].meshes.append(int(me))
            model.materials = [int(x) for x in z[f"m{mi}_materials"]]
            model.textures = [int(x) for x in z[f"m{mi}_textures"]]
            s.models.append(model)
        return s
