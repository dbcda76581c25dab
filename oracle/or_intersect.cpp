/* or_intersect.cpp -- traceRayEXT equivalent of the CPU oracle (test infrastructure only).
 *
 * Restates what the Vulkan driver does behind shaders/ray_gen.rgen:54-65:
 * closest hit over the TLAS, opaque geometry, cull mask 0xFF, no face culling
 * (source/top_level_acceleration_structure.cpp:28), tMin < t < tMax.
 *
 * Canonical hit rule (BVH independent):
 *   per instance i (in instance order): world-space slab interval [tnI, tfI] of the
 *   instance's padded world box; the ray is moved to object space without
 *   renormalising, so t is the same parameter in both spaces;
 *   per triangle p: watertight edge-function test (Woop, Benthin, Wald 2013) with
 *   explicit fma order, gated by the slab interval of the triangle's own padded box;
 *   accepted when tMin < t < tMax, tnI <= t <= tfI;
 *   the winner is the lexicographic minimum of (t, i, p).
 * Because every BVH box is a union of those padded boxes and the slab arithmetic is
 * monotone in the plane coordinate, a traversal that culls with "tn <= best" can
 * never skip the winner, so brute force and any BVH agree exactly.
 */
#include "or_internal.h"
#include <limits>

void or_ray_space_init(RaySpace& rs, V3 o, V3 d)
{
    rs.o = o; rs.d = d;
    const float eps = 8.271806125530277e-25f; /* 2^-80 */
    for (int k = 0; k < 3; ++k) {
        float dk = comp(d, k);
        float safe = fabsf(dk) > eps ? dk : copysignf(eps, dk);
        float inv = 1.0f / safe;
        comp(rs.invd, k) = inv;
        comp(rs.ood, k) = comp(o, k) * inv;
    }
    int kz = 0;
    if (fabsf(d.y) > fabsf(comp(d, kz))) kz = 1;
    if (fabsf(d.z) > fabsf(comp(d, kz))) kz = 2;
    int kx = (kz + 1) % 3, ky = (kx + 1) % 3;
    if (comp(d, kz) < 0.0f) { int t = kx; kx = ky; ky = t; }
    rs.kx = kx; rs.ky = ky; rs.kz = kz;
    /* a usable dominant component is required; the shear uses the reciprocal, not two more divisions */
    float dz = comp(d, kz);
    rs.valid = fabsf(dz) > eps;
    rs.Sz = 1.0f / dz;
    rs.Sx = comp(d, kx) * rs.Sz;
    rs.Sy = comp(d, ky) * rs.Sz;
}

void or_slab(const RaySpace& rs, const Box& b, float& tn, float& tf)
{
    float n[3], f[3];
    for (int k = 0; k < 3; ++k) {
        float inv = comp(rs.invd, k), ood = comp(rs.ood, k);
        float lo = comp(b.lo, k), hi = comp(b.hi, k);
        float pn = inv >= 0.0f ? lo : hi;
        float pf = inv >= 0.0f ? hi : lo;
        n[k] = or_fma(pn, inv, -ood);
        f[k] = or_fma(pf, inv, -ood);
    }
    tn = or_max(or_max(n[0], n[1]), n[2]);
    tf = or_min(or_min(f[0], f[1]), f[2]);
}

bool or_tri_test(const RaySpace& rs, V3 v0, V3 v1, V3 v2, float delta, float& t, float& u, float& v)
{
    V3 A { v0.x - rs.o.x, v0.y - rs.o.y, v0.z - rs.o.z };
    V3 B { v1.x - rs.o.x, v1.y - rs.o.y, v1.z - rs.o.z };
    V3 C { v2.x - rs.o.x, v2.y - rs.o.y, v2.z - rs.o.z };
    float Akz = comp(A, rs.kz), Bkz = comp(B, rs.kz), Ckz = comp(C, rs.kz);
    float Ax = or_fma(-rs.Sx, Akz, comp(A, rs.kx)), Ay = or_fma(-rs.Sy, Akz, comp(A, rs.ky));
    float Bx = or_fma(-rs.Sx, Bkz, comp(B, rs.kx)), By = or_fma(-rs.Sy, Bkz, comp(B, rs.ky));
    float Cx = or_fma(-rs.Sx, Ckz, comp(C, rs.kx)), Cy = or_fma(-rs.Sy, Ckz, comp(C, rs.ky));
    float U = or_fma(Cx, By, -(Cy * Bx));
    float V = or_fma(Ax, Cy, -(Ay * Cx));
    float W = or_fma(Bx, Ay, -(By * Ax));
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
    float det = (U + V) + W;
    if (det == 0.0f) return false;
    float Az = rs.Sz * Akz, Bz = rs.Sz * Bkz, Cz = rs.Sz * Ckz;
    float T = or_fma(W, Cz, or_fma(V, Bz, U * Az));
    float rcp = 1.0f / det;
    t = T * rcp; u = V * rcp; v = W * rcp;
    Box pb;
    pb.lo = V3 { or_min(or_min(v0.x, v1.x), v2.x) - delta, or_min(or_min(v0.y, v1.y), v2.y) - delta, or_min(or_min(v0.z, v1.z), v2.z) - delta };
    pb.hi = V3 { or_max(or_max(v0.x, v1.x), v2.x) + delta, or_max(or_max(v0.y, v1.y), v2.y) + delta, or_max(or_max(v0.z, v1.z), v2.z) + delta };
    float tn, tf;
    or_slab(rs, pb, tn, tf);
    return t >= tn && t <= tf;
}

static inline bool closer(float t, uint32_t inst, uint32_t prim, const HitRec& best)
{
    if (!best.hit) return true;
    if (t < best.t) return true;
    if (t > best.t) return false;
    if (inst != best.inst) return inst < best.inst;
    return prim < best.prim;
}

static inline Box node_box(const float lo[3], const float hi[3])
{
    return Box { V3 { lo[0], lo[1], lo[2] }, V3 { hi[0], hi[1], hi[2] } };
}

/* One instance: brute force over all triangles or BLAS traversal. */
static void instance_hits(const OrInst& I, uint32_t inst_index, V3 o, V3 d, const RaySpace& ws,
                          float tmin, float tmax, bool use_bvh, HitRec& best, OrStats* st)
{
    float tnI, tfI;
    or_slab(ws, I.box, tnI, tfI);
    if (st) st->box_tests++;
    /* interval of this instance: (tmin, tmax) strict, [tnI, tfI] inclusive */
    if (!(tnI <= tfI) || tfI < tmin || tnI > tmax) return;
    if (best.hit && tnI > best.t) return;
    RaySpace os;
    or_ray_space_init(os, or_xform_point(I.w2o, o), or_xform_dir(I.w2o, d));
    if (!os.valid) return;
    const OrBlas* b = I.blas;
    auto test_prim = [&](uint32_t prim) {
        V3 a, bb, c; b->tri(prim, a, bb, c);
        float t, u, v;
        if (st) st->tri_tests++;
        if (!or_tri_test(os, a, bb, c, b->delta, t, u, v)) return;
        if (!(t > tmin && t < tmax && t >= tnI && t <= tfI)) return;
        if (closer(t, inst_index, prim, best)) { best.hit = true; best.t = t; best.inst = inst_index; best.prim = prim; best.u = u; best.v = v; }
    };
    if (!use_bvh) {
        for (uint32_t p = 0; p < b->prim_count; ++p) test_prim(p);
        return;
    }
    if (b->nodes.empty()) return;
    int32_t stack[96]; int sp = 0;
    stack[sp++] = 0;
    while (sp) {
        int32_t ref = stack[--sp];
        if (ref < 0) {
            uint32_t code = (uint32_t)~ref, first = code >> 3, cnt = (code & 7u) + 1;
            for (uint32_t k = 0; k < cnt; ++k) test_prim(b->leaf_prims[first + k]);
            continue;
        }
        const OrBvhNode& nd = b->nodes[ref];
        float n0, f0, n1, f1;
        or_slab(os, node_box(nd.lo0, nd.hi0), n0, f0);
        or_slab(os, node_box(nd.lo1, nd.hi1), n1, f1);
        if (st) st->box_tests += 2;
        float lo = or_max(tmin, tnI), hi = or_min(tmax, tfI);
        if (best.hit) hi = or_min(hi, best.t);
        bool h0 = nd.c0 != std::numeric_limits<int32_t>::min() && or_max(n0, lo) <= or_min(f0, hi);
        bool h1 = nd.c1 != std::numeric_limits<int32_t>::min() && or_max(n1, lo) <= or_min(f1, hi);
        if (h0 && h1) {
            /* nearer child is popped first */
            if (n1 < n0) { stack[sp++] = nd.c0; stack[sp++] = nd.c1; }
            else { stack[sp++] = nd.c1; stack[sp++] = nd.c0; }
        } else if (h0) stack[sp++] = nd.c0;
        else if (h1) stack[sp++] = nd.c1;
    }
}

void or_closest_hit(const OrTlas* tlas, V3 o, V3 d, float tmin, float tmax, bool use_bvh, HitRec& best, OrStats* st)
{
    best.hit = false; best.t = tmax; best.inst = 0xFFFFFFFFu; best.prim = 0xFFFFFFFFu; best.u = best.v = 0.0f;
    if (st) st->rays++;
    RaySpace ws;
    or_ray_space_init(ws, o, d);
    const uint32_t n = (uint32_t)tlas->inst.size();
    if (!use_bvh) {
        for (uint32_t i = 0; i < n; ++i) instance_hits(tlas->inst[i], i, o, d, ws, tmin, tmax, false, best, nullptr);
        if (st && best.hit) st->hits++;
        return;
    }
    if (tlas->nodes.empty()) return;
    int32_t stack[64]; int sp = 0;
    stack[sp++] = 0;
    while (sp) {
        int32_t ref = stack[--sp];
        if (ref < 0) {
            uint32_t pos = ((uint32_t)~ref) >> 3;
            uint32_t i = tlas->leaf_inst[pos];
            instance_hits(tlas->inst[i], i, o, d, ws, tmin, tmax, true, best, st);
            continue;
        }
        const OrBvhNode& nd = tlas->nodes[ref];
        float n0, f0, n1, f1;
        or_slab(ws, node_box(nd.lo0, nd.hi0), n0, f0);
        or_slab(ws, node_box(nd.lo1, nd.hi1), n1, f1);
        if (st) st->box_tests += 2;
        float hi = best.hit ? or_min(tmax, best.t) : tmax;
        bool h0 = nd.c0 != std::numeric_limits<int32_t>::min() && or_max(n0, tmin) <= or_min(f0, hi);
        bool h1 = nd.c1 != std::numeric_limits<int32_t>::min() && or_max(n1, tmin) <= or_min(f1, hi);
        if (h0 && h1) {
            if (n1 < n0) { stack[sp++] = nd.c0; stack[sp++] = nd.c1; }
            else { stack[sp++] = nd.c1; stack[sp++] = nd.c0; }
        } else if (h0) stack[sp++] = nd.c0;
        else if (h1) stack[sp++] = nd.c1;
    }
    if (st && best.hit) st->hits++;
}

extern "C" void or_intersect(const OrTlas* tlas, uint32_t n, const float* origins, const float* dirs,
                             float tmin, float tmax, int mode, OrHit* hits, float* t_out, OrStats* stats)
{
    for (uint32_t r = 0; r < n; ++r) {
        HitRec h;
        or_closest_hit(tlas, V3 { origins[3 * r], origins[3 * r + 1], origins[3 * r + 2] },
                       V3 { dirs[3 * r], dirs[3 * r + 1], dirs[3 * r + 2] }, tmin, tmax, mode != 0, h, stats);
        hits[r].instance = h.hit ? h.inst : 0xFFFFFFFFu;
        hits[r].primitive = h.hit ? h.prim : 0xFFFFFFFFu;
        hits[r].u = h.u; hits[r].v = h.v;
        if (t_out) t_out[r] = h.hit ? h.t : tmax;
    }
}

/* per-ray traversal work (box tests, triangle tests) for analysing wave-level divergence */
extern "C" void or_intersect_counts(const OrTlas* tlas, uint32_t n, const float* origins, const float* dirs,
                                    float tmin, float tmax, uint32_t* box_tests, uint32_t* tri_tests)
{
    for (uint32_t r = 0; r < n; ++r) {
        HitRec h; OrStats st; memset(&st, 0, sizeof st);
        or_closest_hit(tlas, V3 { origins[3 * r], origins[3 * r + 1], origins[3 * r + 2] },
                       V3 { dirs[3 * r], dirs[3 * r + 1], dirs[3 * r + 2] }, tmin, tmax, true, h, &st);
        box_tests[r] = (uint32_t)st.box_tests; tri_tests[r] = (uint32_t)st.tri_tests;
    }
}
