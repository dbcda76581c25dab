/* or_internal.h -- private helpers of the CPU oracle (test infrastructure only). */
#ifndef OR_INTERNAL_H
#define OR_INTERNAL_H

#include "pt_oracle.h"
#include <cmath>
#include <cstring>
#include <vector>

struct V3 { float x, y, z; };

static inline float or_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
static inline float or_min(float a, float b) { return a < b ? a : b; }
static inline float or_max(float a, float b) { return a > b ? a : b; }
static inline float& comp(V3& v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }
static inline float comp(const V3& v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }

/* normalize as x * (1 / sqrt(x.x)) with the dot product accumulated x, then y, then z */
static inline V3 or_normalize(V3 v)
{
    float len2 = or_fma(v.z, v.z, or_fma(v.y, v.y, v.x * v.x));
    float inv = 1.0f / sqrtf(len2);
    return V3 { v.x * inv, v.y * inv, v.z * inv };
}

/* rows r[0..2][0..3] of a 3x4 affine matrix applied to a point / a direction */
static inline V3 or_xform_point(const float m[12], V3 p)
{
    return V3 {
        or_fma(m[0], p.x, or_fma(m[1], p.y, or_fma(m[2], p.z, m[3]))),
        or_fma(m[4], p.x, or_fma(m[5], p.y, or_fma(m[6], p.z, m[7]))),
        or_fma(m[8], p.x, or_fma(m[9], p.y, or_fma(m[10], p.z, m[11]))) };
}
static inline V3 or_xform_dir(const float m[12], V3 d)
{
    return V3 {
        or_fma(m[0], d.x, or_fma(m[1], d.y, m[2] * d.z)),
        or_fma(m[4], d.x, or_fma(m[5], d.y, m[6] * d.z)),
        or_fma(m[8], d.x, or_fma(m[9], d.y, m[10] * d.z)) };
}

struct Box { V3 lo, hi; };

/* Per-space ray with everything the gated triangle test and the slab test need. */
struct RaySpace {
    V3 o, d;
    V3 invd, ood;      /* slab test: t(x) = fma(x, invd, -ood) */
    int kx, ky, kz;    /* watertight test axis permutation */
    float Sx, Sy, Sz;
    bool valid;
};

void or_ray_space_init(RaySpace& rs, V3 o, V3 d);
/* slab interval of a box in this ray space; returns tn, tf (no clipping). */
void or_slab(const RaySpace& rs, const Box& b, float& tn, float& tf);
/* gated watertight ray/triangle test; true when the edge tests, det, and the
 * padded-box gate pass.  Interval and closest-hit ordering are the caller's. */
bool or_tri_test(const RaySpace& rs, V3 v0, V3 v1, V3 v2, float delta, float& t, float& u, float& v);

struct OrBlas {
    const uint8_t* vertices; uint32_t stride; uint32_t max_vertex;
    const uint32_t* indices; uint32_t prim_count; uint32_t max_leaf;
    float delta; Box root;
    std::vector<OrBvhNode> nodes;
    std::vector<uint32_t> leaf_prims;
    std::vector<uint64_t> sorted_keys;
    V3 vertex_pos(uint32_t vi) const { V3 p; memcpy(&p, vertices + (size_t)vi * stride, 12); return p; }
    void tri(uint32_t prim, V3& a, V3& b, V3& c) const
    {
        a = vertex_pos(indices[prim * 3 + 0]); b = vertex_pos(indices[prim * 3 + 1]); c = vertex_pos(indices[prim * 3 + 2]);
    }
};

struct OrInst {
    float o2w[12], w2o[12];
    Box box;            /* padded world box */
    uint32_t custom_index;
    const OrBlas* blas;
};

struct OrTlas {
    std::vector<OrInst> inst;
    std::vector<OrBvhNode> nodes;
    std::vector<uint32_t> leaf_inst;
};

struct OrScene {
    const OrTlas* tlas;
    std::vector<OrMaterial> materials;
    std::vector<OrGeometryNode> geometry_nodes;
    std::vector<OrBLASInstance> blas_instances;
    std::vector<OrTexture> textures;
};

/* canonical LBVH builder shared by BLAS (triangles) and TLAS (instances) */
void or_build_lbvh(const std::vector<Box>& prim_boxes, uint32_t max_leaf,
                   std::vector<OrBvhNode>& nodes, std::vector<uint32_t>& leaf_prims,
                   std::vector<uint64_t>* sorted_keys);

struct HitRec { float t; uint32_t inst, prim; float u, v; bool hit; };
void or_closest_hit(const OrTlas* tlas, V3 o, V3 d, float tmin, float tmax, bool use_bvh, HitRec& out, OrStats* st);

#endif
