"""Binary layouts shared by the host side, the C-ABI library and the tests.

These are the reference's device-visible structs, byte for byte:
  Vertex        include/model_loader.hpp:36-41 == shaders/closest_hit.rchit:12-17 (scalar layout, 32 B)
  Material      shaders/bindless.glsl:3-27 (std140, 96 B) == include/resources/gpu_resources.hpp:157-183
  GeometryNode  shaders/bindless.glsl:33-38 == gpu_resources.hpp:192-200 (32 B)
  BLASInstance  shaders/bindless.glsl:44-47 (4 B)
  Instance      VkAccelerationStructureInstanceKHR as filled by
                source/top_level_acceleration_structure.cpp:25-36 (64 B)
"""
import numpy as np

NULL_RESOURCE_INDEX_VALUE = 0xFFFF  # include/resources/resource_manager.hpp:6
MAX_RESOURCES = 1024                # include/resources/bindless_resources.hpp:65

VERTEX = np.dtype([("pos", "<f4", 3), ("normal", "<f4", 3), ("uv", "<f4", 2)])

MATERIAL = np.dtype([
    ("albedoFactor", "<f4", 4),
    ("metallicFactor", "<f4"), ("roughnessFactor", "<f4"), ("normalScale", "<f4"), ("occlusionStrength", "<f4"),
    ("emissiveFactor", "<f4", 3), ("useEmissiveMap", "<i4"),
    ("useAlbedoMap", "<i4"), ("useMRMap", "<i4"), ("useNormalMap", "<i4"), ("useOcclusionMap", "<i4"),
    ("albedoMapIndex", "<u4"), ("mrMapIndex", "<u4"), ("normalMapIndex", "<u4"), ("occlusionMapIndex", "<u4"),
    ("emissiveMapIndex", "<u4"), ("_pad", "<u4", 3),
])

GEOMETRY_NODE = np.dtype([
    ("vertexBufferDeviceAddress", "<u8"), ("indexBufferDeviceAddress", "<u8"),
    ("materialIndex", "<u4"), ("_pad", "<u4", 3),
])

BLAS_INSTANCE = np.dtype([("firstGeometryIndex", "<u4")])

INSTANCE = np.dtype([
    ("transform", "<f4", 12),
    ("customIndexAndMask", "<u4"),
    ("sbtOffsetAndFlags", "<u4"),
    ("blasRef", "<u8"),
])

TEXTURE_DESC = np.dtype([("rgba8", "<u8"), ("width", "<u4"), ("height", "<u4")])

# 64-byte 4-wide node of the traversal layout (pt_blas_wide_download)
WIDE_NODE = np.dtype([("origin", "<f4", 3), ("scale_x", "<f4"), ("child", "<i4", 4), ("q", "<u4", 6), ("scale_y", "<f4"), ("scale_z", "<f4")])

BVH_NODE = np.dtype([
    ("lo0", "<f4", 3), ("hi0", "<f4", 3), ("lo1", "<f4", 3), ("hi1", "<f4", 3),
    ("c0", "<i4"), ("c1", "<i4"), ("aux0", "<u4"), ("aux1", "<u4"),
])

HIT = np.dtype([("instance", "<u4"), ("primitive", "<u4"), ("u", "<f4"), ("v", "<f4")])

assert VERTEX.itemsize == 32 and MATERIAL.itemsize == 96 and GEOMETRY_NODE.itemsize == 32
assert INSTANCE.itemsize == 64 and BVH_NODE.itemsize == 64 and HIT.itemsize == 16 and TEXTURE_DESC.itemsize == 16

VK_GEOMETRY_INSTANCE_TRIANGLE_FACING_CULL_DISABLE_BIT_KHR = 0x1


def default_material():
    """MaterialCreation defaults copied through Material::Material
    (include/resources/gpu_resources.hpp:112-133, source/resources/gpu_resources.cpp:414-434)."""
    m = np.zeros((), dtype=MATERIAL)
    m["albedoFactor"] = (1.0, 1.0, 1.0, 1.0)
    for k in ("albedoMapIndex", "mrMapIndex", "normalMapIndex", "occlusionMapIndex", "emissiveMapIndex"):
        m[k] = NULL_RESOURCE_INDEX_VALUE
    return m


def make_instance(world3x4, custom_index, blas_ref):
    """One TLAS instance record as top_level_acceleration_structure.cpp:25-36 fills it."""
    r = np.zeros((), dtype=INSTANCE)
    r["transform"] = np.asarray(world3x4, dtype=np.float32).reshape(12)
    r["customIndexAndMask"] = (custom_index & 0xFFFFFF) | (0xFF << 24)
    r["sbtOffsetAndFlags"] = 0 | (VK_GEOMETRY_INSTANCE_TRIANGLE_FACING_CULL_DISABLE_BIT_KHR << 24)
    r["blasRef"] = blas_ref
    return r
