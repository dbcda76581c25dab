// pt_device.h -- device-side arithmetic of the path tracer: ray spaces, the slab test, the gated
// watertight triangle test, two-level BVH traversal, sampling.
//
// Arithmetic contract (results must be bit-identical to an independent CPU evaluation of the same
// formulas): binary32, no contraction (the library is built with -ffp-contract=off), every fused
// operation is an explicit __builtin_fmaf, division and sqrt are the correctly rounded ones.
#pragma once
#include "pt_internal.h"

#define PT_FMA(a, b, c) __builtin_fmaf((a), (b), (c))

struct PtRaySpace {
    float ox, oy, oz, dx, dy, dz;
    float ix, iy, iz;        // 1 / d with |d| clamped away from zero
    float oox, ooy, ooz;     // o * (1 / d)
};

__device__ __forceinline__ float pt_safe_rcp(float d)
{
    const float eps = 8.271806125530277e-25f; // 2^-80
    float s = fabsf(d) > eps ? d : copysignf(eps, d);
    return 1.0f / s;
}

__device__ __forceinline__ void pt_ray_space_init(PtRaySpace& r, float ox, float oy, float oz, float dx, float dy, float dz)
{
    r.ox = ox; r.oy = oy; r.oz = oz; r.dx = dx; r.dy = dy; r.dz = dz;
    r.ix = pt_safe_rcp(dx); r.iy = pt_safe_rcp(dy); r.iz = pt_safe_rcp(dz);
    r.oox = ox * r.ix; r.ooy = oy * r.iy; r.ooz = oz * r.iz;
}

// Slab interval of a box: plane distance t(x) = fma(x, 1/d, -(o/d)).  The function is monotone in x,
// so an enclosing box never yields a narrower interval than the boxes it contains.
__device__ __forceinline__ void pt_slab(const PtRaySpace& r, float lox, float loy, float loz, float hix, float hiy, float hiz,
                                        float& tn, float& tf)
{
    float ax = PT_FMA(lox, r.ix, -r.oox), bx = PT_FMA(hix, r.ix, -r.oox);
    float ay = PT_FMA(loy, r.iy, -r.ooy), by = PT_FMA(hiy, r.iy, -r.ooy);
    float az = PT_FMA(loz, r.iz, -r.ooz), bz = PT_FMA(hiz, r.iz, -r.ooz);
    tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
}

// Axis permutation and shear of the watertight ray/triangle test (Woop, Benthin, Wald 2013).
struct PtWoop { int kx, ky, kz; float Sx, Sy, Sz; };

__device__ __forceinline__ float pt_sel3(int k, float x, float y, float z) { return k == 0 ? x : (k == 1 ? y : z); }

__device__ __forceinline__ bool pt_woop_init(PtWoop& w, float dx, float dy, float dz)
{
    int kz = 0;
    float m = fabsf(dx);
    if (fabsf(dy) > m) { kz = 1; m = fabsf(dy); }
    if (fabsf(dz) > m) { kz = 2; }
    int kx = kz == 2 ? 0 : kz + 1;
    int ky = kx == 2 ? 0 : kx + 1;
    float dkz = pt_sel3(kz, dx, dy, dz);
    if (dkz < 0.0f) { int t = kx; kx = ky; ky = t; }
    w.kx = kx; w.ky = ky; w.kz = kz;
    w.Sx = pt_sel3(kx, dx, dy, dz) / dkz;
    w.Sy = pt_sel3(ky, dx, dy, dz) / dkz;
    w.Sz = 1.0f / dkz;
    return dkz != 0.0f;
}

// Gated watertight test.  Returns true when the edge functions agree in sign, det != 0 and t lies
// inside the slab interval of the triangle's own box padded by delta.
__device__ __forceinline__ bool pt_tri_test(const PtRaySpace& r, const PtWoop& w, float delta,
                                            float v0x, float v0y, float v0z, float v1x, float v1y, float v1z,
                                            float v2x, float v2y, float v2z, float& t, float& u, float& v)
{
    float Ax_ = v0x - r.ox, Ay_ = v0y - r.oy, Az_ = v0z - r.oz;
    float Bx_ = v1x - r.ox, By_ = v1y - r.oy, Bz_ = v1z - r.oz;
    float Cx_ = v2x - r.ox, Cy_ = v2y - r.oy, Cz_ = v2z - r.oz;
    float Akz = pt_sel3(w.kz, Ax_, Ay_, Az_), Bkz = pt_sel3(w.kz, Bx_, By_, Bz_), Ckz = pt_sel3(w.kz, Cx_, Cy_, Cz_);
    float Ax = PT_FMA(-w.Sx, Akz, pt_sel3(w.kx, Ax_, Ay_, Az_)), Ay = PT_FMA(-w.Sy, Akz, pt_sel3(w.ky, Ax_, Ay_, Az_));
    float Bx = PT_FMA(-w.Sx, Bkz, pt_sel3(w.kx, Bx_, By_, Bz_)), By = PT_FMA(-w.Sy, Bkz, pt_sel3(w.ky, Bx_, By_, Bz_));
    float Cx = PT_FMA(-w.Sx, Ckz, pt_sel3(w.kx, Cx_, Cy_, Cz_)), Cy = PT_FMA(-w.Sy, Ckz, pt_sel3(w.ky, Cx_, Cy_, Cz_));
    float U = PT_FMA(Cx, By, -(Cy * Bx));
    float V = PT_FMA(Ax, Cy, -(Ay * Cx));
    float W = PT_FMA(Bx, Ay, -(By * Ax));
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
    float det = (U + V) + W;
    if (det == 0.0f) return false;
    float Az = w.Sz * Akz, Bz = w.Sz * Bkz, Cz = w.Sz * Ckz;
    float T = PT_FMA(W, Cz, PT_FMA(V, Bz, U * Az));
    float rcp = 1.0f / det;
    t = T * rcp; u = V * rcp; v = W * rcp;
    float tn, tf;
    pt_slab(r, fminf(fminf(v0x, v1x), v2x) - delta, fminf(fminf(v0y, v1y), v2y) - delta, fminf(fminf(v0z, v1z), v2z) - delta,
            fmaxf(fmaxf(v0x, v1x), v2x) + delta, fmaxf(fmaxf(v0y, v1y), v2y) + delta, fmaxf(fmaxf(v0z, v1z), v2z) + delta, tn, tf);
    return t >= tn && t <= tf;
}

struct PtHitRec { float t; uint32_t inst, prim; float u, v; };

struct PtTlasView {
    const float4* nodes;          // 4 float4 per node
    const uint32_t* leaf_inst;
    const PtDevInstance* inst;
};

#define PT_NODE_SENTINEL 0x7FFFFFFF
#define PT_NODE_DONE 0x7FFFFFFE
#define PT_NODE_EMPTY ((int)0x80000000)

// Per-lane traversal stack: the first PT_LDS_STACK entries live in LDS ([entry][thread], one bank
// per lane), deeper entries spill to a per-thread slice of a global buffer.
#define PT_LDS_STACK 24
#define PT_SPILL_STACK 104

struct PtStack {
    int* lds;          // &s_stack[0][threadIdx.x]
    int* spill;        // per-thread slice of PT_SPILL_STACK ints
    int sp;
    int lds_stride;    // blockDim.x
    __device__ __forceinline__ void push(int v)
    {
        if (sp < PT_LDS_STACK) lds[sp * lds_stride] = v;
        else spill[sp - PT_LDS_STACK] = v;
        ++sp;
    }
    __device__ __forceinline__ int pop()
    {
        if (sp == 0) return PT_NODE_DONE;
        --sp;
        return sp < PT_LDS_STACK ? lds[sp * lds_stride] : spill[sp - PT_LDS_STACK];
    }
};

// traceRayEXT(topLevelAS, gl_RayFlagsOpaqueEXT, 0xFF, 0, 0, 0, o, tmin, d, tmax, 0) of
// shaders/ray_gen.rgen:54-65 with face culling disabled per instance
// (source/top_level_acceleration_structure.cpp:28).
// Closest hit = lexicographic minimum of (t, instance, primitive) over every triangle passing the
// gated test with tmin < t < tmax and t inside the instance's world-box interval.
__device__ __forceinline__ void pt_closest_hit(const PtTlasView& tl, PtStack& stk, float ox, float oy, float oz,
                                               float dx, float dy, float dz, float tmin, float tmax, PtHitRec& best)
{
    best.t = tmax; best.inst = 0xFFFFFFFFu; best.prim = 0xFFFFFFFFu; best.u = 0.0f; best.v = 0.0f;
    PtRaySpace ws, cur;
    pt_ray_space_init(ws, ox, oy, oz, dx, dy, dz);
    cur = ws;
    PtWoop woop; woop.kx = 0; woop.ky = 1; woop.kz = 2; woop.Sx = woop.Sy = woop.Sz = 0.0f;
    const float4* nodes = tl.nodes;
    const float4* tris = nullptr;
    float delta = 0.0f;
    float tlo = tmin, thi = tmax;   // closed interval handed to box tests in the current space
    float tnI = 0.0f, tfI = 0.0f;
    uint32_t cur_inst = 0;
    bool in_blas = false;
    stk.sp = 0;
    int node = 0;
    for (;;) {
        while ((unsigned)node < (unsigned)PT_NODE_DONE) {
            const float4 n0 = nodes[4 * (size_t)node + 0], n1 = nodes[4 * (size_t)node + 1];
            const float4 n2 = nodes[4 * (size_t)node + 2], n3 = nodes[4 * (size_t)node + 3];
            const int c0 = __float_as_int(n3.x), c1 = __float_as_int(n3.y);
            float tn0, tf0, tn1, tf1;
            pt_slab(cur, n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, tn0, tf0);
            pt_slab(cur, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, tn1, tf1);
            const float hi = fminf(thi, best.t);
            const bool h0 = fmaxf(tn0, tlo) <= fminf(tf0, hi);
            const bool h1 = (c1 != PT_NODE_EMPTY) && fmaxf(tn1, tlo) <= fminf(tf1, hi);
            if (h0 && h1) {
                const bool swap = tn1 < tn0;
                stk.push(swap ? c0 : c1);
                node = swap ? c1 : c0;
            } else if (h0) node = c0;
            else if (h1) node = c1;
            else node = stk.pop();
        }
        if (node == PT_NODE_DONE) break;
        if (node == PT_NODE_SENTINEL) {      // all of the instance's BLAS visited: back to world space
            in_blas = false; cur = ws; nodes = tl.nodes; tlo = tmin; thi = tmax;
            node = stk.pop();
            continue;
        }
        const uint32_t code = ~(uint32_t)node;
        const uint32_t first = code >> 3, cnt = (code & 7u) + 1u;
        if (!in_blas) {
            const uint32_t ii = tl.leaf_inst[first];
            const float4* q = tl.inst[ii].q;
            const float4 blo = q[3], bhi = q[4];
            float a, b;
            pt_slab(ws, blo.x, blo.y, blo.z, bhi.x, bhi.y, bhi.z, a, b);
            bool enter = (a <= b) && !(b < tmin) && !(a > tmax) && !(a > best.t);
            if (enter) {
                const float4 r0 = q[0], r1 = q[1], r2 = q[2];
                float oox = PT_FMA(r0.x, ws.ox, PT_FMA(r0.y, ws.oy, PT_FMA(r0.z, ws.oz, r0.w)));
                float ooy = PT_FMA(r1.x, ws.ox, PT_FMA(r1.y, ws.oy, PT_FMA(r1.z, ws.oz, r1.w)));
                float ooz = PT_FMA(r2.x, ws.ox, PT_FMA(r2.y, ws.oy, PT_FMA(r2.z, ws.oz, r2.w)));
                float odx = PT_FMA(r0.x, ws.dx, PT_FMA(r0.y, ws.dy, r0.z * ws.dz));
                float ody = PT_FMA(r1.x, ws.dx, PT_FMA(r1.y, ws.dy, r1.z * ws.dz));
                float odz = PT_FMA(r2.x, ws.dx, PT_FMA(r2.y, ws.dy, r2.z * ws.dz));
                enter = pt_woop_init(woop, odx, ody, odz);
                if (enter) {
                    pt_ray_space_init(cur, oox, ooy, ooz, odx, ody, odz);
                    const float4 p = q[5];
                    nodes = (const float4*)(((uint64_t)__float_as_uint(p.y) << 32) | __float_as_uint(p.x));
                    tris = (const float4*)(((uint64_t)__float_as_uint(p.w) << 32) | __float_as_uint(p.z));
                    delta = blo.w;
                    tnI = a; tfI = b; cur_inst = ii;
                    tlo = fmaxf(tmin, a); thi = fminf(tmax, b);
                    in_blas = true;
                    stk.push(PT_NODE_SENTINEL);
                    node = 0;
                    continue;
                }
            }
            node = stk.pop();
            continue;
        }
        for (uint32_t k = 0; k < cnt; ++k) {
            const float4 a = tris[3 * (size_t)(first + k) + 0], b = tris[3 * (size_t)(first + k) + 1], c = tris[3 * (size_t)(first + k) + 2];
            float t, u, v;
            if (pt_tri_test(cur, woop, delta, a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, t, u, v)) {
                const uint32_t prim = __float_as_uint(c.y);
                if (t > tmin && t < tmax && t >= tnI && t <= tfI) {
                    const bool closer = (t < best.t) || (t == best.t && (cur_inst < best.inst || (cur_inst == best.inst && prim < best.prim)));
                    if (closer) { best.t = t; best.inst = cur_inst; best.prim = prim; best.u = u; best.v = v; }
                }
            }
        }
        node = stk.pop();
    }
}

// ---- shaders/sampling.glsl ----------------------------------------------------------------
__device__ __forceinline__ uint32_t pt_tea(uint32_t val0, uint32_t val1)   // :4-18
{
    uint32_t v0 = val0, v1 = val1, s0 = 0;
#pragma unroll
    for (uint32_t n = 0; n < 16; ++n) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}
__device__ __forceinline__ float pt_random(uint32_t& prev)                   // :22-34
{
    prev = 1664525u * prev + 1013904223u;
    return (float)(prev & 0x00FFFFFFu) / (float)0x01000000;
}
// Pinned sin/cos for x >= 0: Cody-Waite reduction by pi/2 in three parts, then the classic
// single-precision minimax polynomials on [-pi/4, pi/4].
__device__ __forceinline__ void pt_sincos(float x, float& s_out, float& c_out)
{
    int k = (int)(x * 0.636619772f + 0.5f);
    float kf = (float)k;
    float r = PT_FMA(kf, -1.5703125f, x);
    r = PT_FMA(kf, -4.837512969970703125e-4f, r);
    r = PT_FMA(kf, -7.54978995489188216e-8f, r);
    float r2 = r * r;
    float ps = PT_FMA(PT_FMA(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f);
    float s = PT_FMA(r * r2, ps, r);
    float pc = PT_FMA(PT_FMA(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f);
    float c = PT_FMA(r2 * r2, pc, PT_FMA(-0.5f, r2, 1.0f));
    int q = k & 3;
    s_out = q == 0 ? s : (q == 1 ? c : (q == 2 ? -s : -c));
    c_out = q == 0 ? c : (q == 1 ? -s : (q == 2 ? -c : s));
}
__device__ __forceinline__ void pt_normalize(float& x, float& y, float& z)
{
    float len2 = PT_FMA(z, z, PT_FMA(y, y, x * x));
    float inv = 1.0f / sqrtf(len2);
    x *= inv; y *= inv; z *= inv;
}
