/*
 * pt_oracle.h -- CPU restatement of the reference path tracer's hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product: the HIP library never links, loads or calls it.  Only the tests,
 * the smoke check and the CPU-baseline leg of the benchmark use it, and only
 * as the checker / the thing timed as "CPU baseline".
 *
 * What is restated (paths relative to the reference checkout):
 *   shaders/sampling.glsl:4-65      TEA, LCG, Random, SamplingHemisphere,
 *                                   CreateCoordinateSystem
 *   shaders/ray.glsl:1-9            HitPayload
 *   shaders/ray_gen.rgen:19-88      primary ray, sample loop, bounce loop,
 *                                   running-mean accumulation
 *   shaders/closest_hit.rchit:33-86 UnpackGeometry + diffuse scatter
 *   shaders/miss.rmiss:8-21         background / ambient + path termination
 *   shaders/bindless.glsl:1-51      Material / GeometryNode / BLASInstance
 *   source/renderer.cpp:172-193     camera matrices (consumed, not built, here)
 *   source/top_level_acceleration_structure.cpp:19-43   instance records
 *   source/bottom_level_acceleration_structure.cpp:34-78 BLAS inputs
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or images for
 * this path, and its ray/triangle test, BVH and sin/cos live in the Vulkan
 * driver.  The oracle is pinned only by known answers derived from the shader
 * text (TEA/LCG values, closed-form primary rays, furnace-test algebra).
 *
 * Arithmetic contract shared with the HIP implementation (written twice,
 * independently, from this description):
 *   - binary32, round-to-nearest-even, no contraction; "fma" below is a single
 *     rounding fused multiply-add; division and sqrt are IEEE correctly rounded.
 *   - sin/cos are a fixed polynomial (or_sincos) instead of the driver's.
 *   - the closest hit is the lexicographic minimum of (t, instance, primitive)
 *     over all (instance, triangle) pairs that pass the gated watertight test,
 *     which makes the result independent of any BVH.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- device-ABI structs (shaders/bindless.glsl, include/resources/gpu_resources.hpp:157-205) */
typedef struct OrVertex { float pos[3]; float normal[3]; float uv[2]; } OrVertex; /* 32 B */

typedef struct OrMaterial {          /* std140, 96 B */
    float    albedoFactor[4];
    float    metallicFactor, roughnessFactor, normalScale, occlusionStrength;
    float    emissiveFactor[3];
    int32_t  useEmissiveMap;
    int32_t  useAlbedoMap, useMRMap, useNormalMap, useOcclusionMap;
    uint32_t albedoMapIndex, mrMapIndex, normalMapIndex, occlusionMapIndex;
    uint32_t emissiveMapIndex;
    uint32_t pad[3];
} OrMaterial;

typedef struct OrGeometryNode {      /* 32 B */
    uint64_t vertexBufferDeviceAddress;
    uint64_t indexBufferDeviceAddress;
    uint32_t materialIndex;
    uint32_t pad[3];
} OrGeometryNode;

typedef struct OrBLASInstance { uint32_t firstGeometryIndex; } OrBLASInstance;

/* VkAccelerationStructureInstanceKHR layout (64 B); blasRef holds an OrBlas*. */
typedef struct OrInstance {
    float    transform[12];          /* row-major 3x4 object->world */
    uint32_t customIndexAndMask;     /* [23:0] instanceCustomIndex, [31:24] mask */
    uint32_t sbtOffsetAndFlags;      /* [23:0] sbt offset, [31:24] flags */
    uint64_t blasRef;
} OrInstance;

typedef struct OrTexture { const uint8_t* rgba8; uint32_t width, height; } OrTexture;

/* Emitted BVH node: two child boxes + two child references (64 B). */
typedef struct OrBvhNode {
    float   lo0[3], hi0[3], lo1[3], hi1[3];
    int32_t c0, c1;                  /* >=0 inner node; <0 leaf: ~c = first<<3 | (count-1); INT32_MIN empty */
    uint32_t aux0, aux1;
} OrBvhNode;

typedef struct OrHit { uint32_t instance, primitive; float u, v; } OrHit; /* instance = 0xFFFFFFFF on miss */

typedef struct OrStats {
    uint64_t rays;        /* traceRay equivalents, terminating miss rays included */
    uint64_t box_tests;   /* child boxes tested (canonical 32-B node visits) */
    uint64_t tri_tests;   /* triangles tested */
    uint64_t hits;        /* rays that hit */
    uint64_t tex_hits;    /* hits on a material with an albedo map */
    uint64_t paths;
} OrStats;

typedef struct OrBlas OrBlas;
typedef struct OrTlas OrTlas;
typedef struct OrScene OrScene;

/* ---- sampling.glsl */
uint32_t or_tea(uint32_t v0, uint32_t v1);
uint32_t or_lcg(uint32_t* state);
float    or_random(uint32_t* state);
void     or_sincos(float x, float* s, float* c);
void     or_create_coordinate_system(const float N[3], float Nt[3], float Nb[3]);
void     or_sampling_hemisphere(uint32_t* seed, const float x[3], const float y[3], const float z[3], float out[3]);

/* ---- ray_gen.rgen:21-27; camera = viewInverse[16] then projInverse[16], column-major */
void or_primary_ray(const float camera[32], uint32_t W, uint32_t H, uint32_t px, uint32_t py, float o[3], float d[3]);
void or_invert_transform(const float o2w[12], float w2o[12]);

/* ---- acceleration structures */
OrBlas* or_blas_build(const void* vertices, uint32_t vertex_stride, uint32_t max_vertex,
                      const uint32_t* indices, uint32_t prim_count, uint32_t max_leaf);
void     or_blas_destroy(OrBlas*);
uint32_t or_blas_node_count(const OrBlas*);
const OrBvhNode* or_blas_nodes(const OrBlas*);
const uint32_t*  or_blas_leaf_prims(const OrBlas*);   /* prim ids in sorted (Morton) order */
const uint64_t*  or_blas_sorted_keys(const OrBlas*);
void     or_blas_root_box(const OrBlas*, float lo[3], float hi[3]);
float    or_blas_delta(const OrBlas*);

OrTlas*  or_tlas_build(const OrInstance* instances, uint32_t n);
void     or_tlas_destroy(OrTlas*);
uint32_t or_tlas_node_count(const OrTlas*);
const OrBvhNode* or_tlas_nodes(const OrTlas*);
const uint32_t*  or_tlas_leaf_instances(const OrTlas*);
void     or_tlas_instance_box(const OrTlas*, uint32_t i, float lo[3], float hi[3]);
void     or_tlas_instance_w2o(const OrTlas*, uint32_t i, float w2o[12]);

/* ---- scene tables (set 0 of the reference's descriptor layout) */
OrScene* or_scene_create(const OrTlas* tlas,
                         const OrMaterial* materials, uint32_t n_mat,
                         const OrGeometryNode* geometry_nodes, uint32_t n_geo,
                         const OrBLASInstance* blas_instances, uint32_t n_inst,
                         const OrTexture* textures, uint32_t n_tex);
void     or_scene_destroy(OrScene*);

/* ---- traceRayEXT equivalent on explicit rays.  mode 0 = brute force, 1 = BVH. */
void or_intersect(const OrTlas* tlas, uint32_t n, const float* origins, const float* dirs,
                  float tmin, float tmax, int mode, OrHit* hits, float* t_out, OrStats* stats);

void or_intersect_counts(const OrTlas* tlas, uint32_t n, const float* origins, const float* dirs,
                         float tmin, float tmax, uint32_t* box_tests, uint32_t* tri_tests);

/* ---- one frame of ray_gen.rgen over the pixel rectangle [x0,x1) x [y0,y1).
 * result: float3 per pixel of the FULL image (row-major, W*H*3), only the rectangle is written.
 * seed for sample i of this frame: TEA(y*W+x, frame*spp + i). */
void or_render_frame(const OrScene* scene, const float camera[32], uint32_t W, uint32_t H,
                     uint32_t frame, uint32_t spp, uint32_t max_depth,
                     uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                     int use_bvh, int threads, float* result, OrStats* stats);

/* ---- per-bounce record of a single path (debugging aid for parity failures) */
typedef struct OrBounce {
    OrHit hit; float t;
    float origin[3], dir[3];      /* the ray traced at this depth */
    float hit_value[3], weight[3];/* payload after the hit / miss shader */
    uint32_t seed_after;
} OrBounce;
uint32_t or_trace_path(const OrScene* scene, const float camera[32], uint32_t W, uint32_t H,
                       uint32_t px, uint32_t py, uint32_t sample_index, uint32_t max_depth,
                       OrBounce* bounces, float radiance[3]);

/* ---- ray_gen.rgen:76-87.  mode 0: float sum (rgb += result, a += 1);
 * mode 1: float running mean (literal mix); mode 2: 8-bit UNORM running mean (literal). */
void or_accumulate(int mode, uint32_t n_pixels, uint32_t frame, const float* result, void* accum);
/* float sum buffer -> mean -> clamp -> RGBA8 */
void or_resolve(uint32_t n_pixels, const float* accum_sum, uint8_t* rgba8);

float or_texture_bilinear_channel(const OrTexture* tex, float u, float v, int channel);

#ifdef __cplusplus
}
#endif
#endif
