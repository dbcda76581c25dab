// pt_device.h -- device-side arithmetic of the path tracer: ray spaces, the slab test, the gated
// watertight triangle test, two-level BVH traversal, sampling.
//
// Arithmetic contract (results must be bit-identical to an independent CPU evaluation of the same
// formulas): binary32, no contraction (the library is built with -ffp-contract=off), every fused
// operation is an explicit __builtin_fmaf, division and sqrt are the correctly rounded ones.
#pragma once
#include "pt_internal.h"

#define PT_FMA(a, b, c) __builtin_fmaf((a), (b), (c))
#define PT_DIR_EPS 8.271806125530277e-25f   // 2^-80

// Queue reads and writes: 16 bytes, 16-byte aligned, touched once, marked non-temporal so that they
// do not displace the BVH in L2 (worth 0.7 % on the dragon frame).
typedef float pt_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 pt_stream_load(const float4* p)
{
    const pt_v4f v = __builtin_nontemporal_load((const pt_v4f*)p);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void pt_stream_store(float4* p, const float4 v)
{
    pt_v4f w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
    __builtin_nontemporal_store(w, (pt_v4f*)p);
}

// 1 / x and sqrt(x) with the bits of the correctly rounded operations, in a third of the instructions of the
// compiler's IEEE expansions (v_div_scale x 2, v_rcp, six fmas, v_div_fmas, v_div_fixup): the hardware
// approximation and one residual correction, r = rcp(x), r += r * fma(-x, r, 1); s = x * rsq(x), s += (rsq(x) / 2) *
// fma(-s, s, x).  tools/micro/rcp_exact.hip compares both with the IEEE forms over every binary32 pattern: not one
// differs for 2^-126 <= |x| <= 2^126 (reciprocal) and 2^-100 <= x <= 2^100 (root).  Outside 2^-100 .. 2^100
// (zeros, subnormals, infinities, NaNs) the whole wave takes the IEEE form, which happens only with rays or
// triangles that are not finite or degenerate.
#define PT_EXACT_LO 0x1p-100f
#define PT_EXACT_HI 0x1p+100f
__device__ __forceinline__ float pt_rcp_core(float x)
{
    const float r = __builtin_amdgcn_rcpf(x);
    return PT_FMA(PT_FMA(-x, r, 1.0f), r, r);
}
__device__ __forceinline__ float pt_rcp(float x)
{
    const float ax = fabsf(x);
    if (__ballot(!(ax >= PT_EXACT_LO && ax <= PT_EXACT_HI)) != 0ull) return 1.0f / x;
    return pt_rcp_core(x);
}
__device__ __forceinline__ float pt_sqrt(float x)
{
    if (__ballot(!(x >= PT_EXACT_LO && x <= PT_EXACT_HI)) != 0ull) return sqrtf(x);
    const float y = __builtin_amdgcn_rsqf(x);
    const float s = x * y, h = 0.5f * y;
    return PT_FMA(PT_FMA(-s, s, x), h, s);
}

// x / c for the two constant divisors of the shaders (pi in closest_hit.rchit:73,80, 255 in the UNORM8 texel):
// q = x * RN(1 / c), q += RN(1 / c) * fma(-c, q, x).  Over every binary32 pattern (tools/micro/rcp_exact.hip) the
// result has the bits of the division for 2^-100 <= |x| <= 2^100 and for +0; a texel byte is one of 256 such values.
__device__ __forceinline__ float pt_div_pi(float x)
{
    constexpr float c = 3.14159265f, rc = 1.0f / c;
    const float ax = fabsf(x);
    if (__ballot(!((ax >= PT_EXACT_LO && ax <= PT_EXACT_HI) || __float_as_uint(x) == 0u)) != 0ull) return x / c;
    const float q = x * rc;
    return PT_FMA(PT_FMA(-c, q, x), rc, q);
}
__device__ __forceinline__ float pt_unorm8(uint32_t byte)
{
    constexpr float c = 255.0f, rc = 1.0f / c;
    const float x = (float)byte, q = x * rc;
    return PT_FMA(PT_FMA(-c, q, x), rc, q);
}

struct PtRaySpace {
    float ox, oy, oz, dx, dy, dz;
    float ix, iy, iz;        // 1 / d with |d| clamped away from zero
    float oox, ooy, ooz;     // o * (1 / d)
};

__device__ __forceinline__ float pt_safe_rcp(float d)
{
    float s = fabsf(d) > PT_DIR_EPS ? d : copysignf(PT_DIR_EPS, d);
    return 1.0f / s;
}
__device__ __forceinline__ float pt_dir_clamp(float d) { return fabsf(d) > PT_DIR_EPS ? d : copysignf(PT_DIR_EPS, d); }

__device__ __forceinline__ void pt_ray_space_init(PtRaySpace& r, float ox, float oy, float oz, float dx, float dy, float dz)
{
    r.ox = ox; r.oy = oy; r.oz = oz; r.dx = dx; r.dy = dy; r.dz = dz;
    // the three divisors are at least 2^-80 in magnitude: one test of the largest decides for all three
    const float sx = pt_dir_clamp(dx), sy = pt_dir_clamp(dy), sz = pt_dir_clamp(dz);
    if (__ballot(!(fmaxf(fmaxf(fabsf(sx), fabsf(sy)), fabsf(sz)) <= PT_EXACT_HI)) != 0ull) {
        r.ix = 1.0f / sx; r.iy = 1.0f / sy; r.iz = 1.0f / sz;
    } else {
        r.ix = pt_rcp_core(sx); r.iy = pt_rcp_core(sy); r.iz = pt_rcp_core(sz);
    }
    r.oox = ox * r.ix; r.ooy = oy * r.iy; r.ooz = oz * r.iz;
}

// A ray can hit triangles only if its dominant direction component is a usable divisor.
__device__ __forceinline__ bool pt_ray_valid(float dx, float dy, float dz)
{
    return fmaxf(fmaxf(fabsf(dx), fabsf(dy)), fabsf(dz)) > PT_DIR_EPS;
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

__device__ __forceinline__ float pt_sel3(int k, float x, float y, float z) { return k == 0 ? x : (k == 1 ? y : z); }

// Gated watertight ray/triangle test (Woop, Benthin, Wald 2013).  The part that depends on the ray alone:
// kz = dominant axis (lowest index on ties), kx and ky the two others in cyclic order, exchanged when d[kz] < 0,
// Sz = 1/d[kz] (the slab test's reciprocal), Sx = d[kx]*Sz, Sy = d[ky]*Sz.
struct PtTriSetup { int kx, ky, kz; float Sx, Sy, Sz; };
__device__ __forceinline__ PtTriSetup pt_tri_setup(float dx, float dy, float dz, float ix, float iy, float iz)
{
    PtTriSetup s;
    s.kz = 0;
    float m = fabsf(dx);
    if (fabsf(dy) > m) { s.kz = 1; m = fabsf(dy); }
    if (fabsf(dz) > m) { s.kz = 2; }
    s.kx = s.kz == 2 ? 0 : s.kz + 1;
    s.ky = s.kx == 2 ? 0 : s.kx + 1;
    if (pt_sel3(s.kz, dx, dy, dz) < 0.0f) { int t = s.kx; s.kx = s.ky; s.ky = t; }
    s.Sz = pt_sel3(s.kz, ix, iy, iz);
    s.Sx = pt_sel3(s.kx, dx, dy, dz) * s.Sz;
    s.Sy = pt_sel3(s.ky, dx, dy, dz) * s.Sz;
    return s;
}
// The test itself, for a ray with origin (ox, oy, oz) whose slab terms are in r.  Returns true when the edge
// functions agree in sign, det != 0 and t lies inside the slab interval of the triangle's own box padded by delta.
__device__ __forceinline__ bool pt_tri_test_with(const PtTriSetup& S, const PtRaySpace& r, float delta,
                                                 float v0x, float v0y, float v0z, float v1x, float v1y, float v1z,
                                                 float v2x, float v2y, float v2z, float& t, float& u, float& v)
{
    const int kx = S.kx, ky = S.ky, kz = S.kz;
    const float Sx = S.Sx, Sy = S.Sy, Sz = S.Sz;
    float Ax_ = v0x - r.ox, Ay_ = v0y - r.oy, Az_ = v0z - r.oz;
    float Bx_ = v1x - r.ox, By_ = v1y - r.oy, Bz_ = v1z - r.oz;
    float Cx_ = v2x - r.ox, Cy_ = v2y - r.oy, Cz_ = v2z - r.oz;
    float Akz = pt_sel3(kz, Ax_, Ay_, Az_), Bkz = pt_sel3(kz, Bx_, By_, Bz_), Ckz = pt_sel3(kz, Cx_, Cy_, Cz_);
    float Ax = PT_FMA(-Sx, Akz, pt_sel3(kx, Ax_, Ay_, Az_)), Ay = PT_FMA(-Sy, Akz, pt_sel3(ky, Ax_, Ay_, Az_));
    float Bx = PT_FMA(-Sx, Bkz, pt_sel3(kx, Bx_, By_, Bz_)), By = PT_FMA(-Sy, Bkz, pt_sel3(ky, Bx_, By_, Bz_));
    float Cx = PT_FMA(-Sx, Ckz, pt_sel3(kx, Cx_, Cy_, Cz_)), Cy = PT_FMA(-Sy, Ckz, pt_sel3(ky, Cx_, Cy_, Cz_));
    float U = PT_FMA(Cx, By, -(Cy * Bx));
    float V = PT_FMA(Ax, Cy, -(Ay * Cx));
    float W = PT_FMA(Bx, Ay, -(By * Ax));
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return false;
    float det = (U + V) + W;
    if (det == 0.0f) return false;
    float Az = Sz * Akz, Bz = Sz * Bkz, Cz = Sz * Ckz;
    float T = PT_FMA(W, Cz, PT_FMA(V, Bz, U * Az));
    float rcp = pt_rcp(det);
    t = T * rcp; u = V * rcp; v = W * rcp;
    float tn, tf;
    pt_slab(r, fminf(fminf(v0x, v1x), v2x) - delta, fminf(fminf(v0y, v1y), v2y) - delta, fminf(fminf(v0z, v1z), v2z) - delta,
            fmaxf(fmaxf(v0x, v1x), v2x) + delta, fmaxf(fmaxf(v0y, v1y), v2y) + delta, fmaxf(fmaxf(v0z, v1z), v2z) + delta, tn, tf);
    return t >= tn && t <= tf;
}
// The form that works the ray's part out on every call (where a lane's registers cannot hold it between calls).
__device__ __forceinline__ bool pt_tri_test(const PtRaySpace& r, float delta,
                                            float v0x, float v0y, float v0z, float v1x, float v1y, float v1z,
                                            float v2x, float v2y, float v2z, float& t, float& u, float& v)
{
    return pt_tri_test_with(pt_tri_setup(r.dx, r.dy, r.dz, r.ix, r.iy, r.iz), r, delta, v0x, v0y, v0z, v1x, v1y, v1z, v2x, v2y, v2z, t, u, v);
}

struct PtHitRec { float t; uint32_t inst, prim; float u, v; };

// Keeps the lexicographic minimum of (t, instance, primitive).  The order is evaluated without
// short-circuits into one predicate and applied with selects: written as nested branches, the
// compiler (ROCm 7.2 clang, gfx950) kept two copies of the record and updated only one of them
// on the tie path, so a tie won by the later candidate returned stale fields.
__device__ __forceinline__ void pt_hit_update(PtHitRec& best, float t, uint32_t inst, uint32_t prim, float u, float v)
{
    const unsigned long long id = ((unsigned long long)inst << 32) | prim;
    const unsigned long long best_id = ((unsigned long long)best.inst << 32) | best.prim;
    const bool closer = (t < best.t) | ((t == best.t) & (id < best_id));
    best.t = closer ? t : best.t;
    best.inst = closer ? inst : best.inst;
    best.prim = closer ? prim : best.prim;
    best.u = closer ? u : best.u;
    best.v = closer ? v : best.v;
}

// One pool per TLAS: [wide nodes of the TLAS | wide nodes of every BLAS] with absolute child
// references, and the triangle records of every BLAS; a node is four float4, a triangle three.
struct PtTlasView {
    const float4* nodes;
    const float4* nodes_f32;       // the same nodes as PtWideF (128 bytes each)
    int n_root;                    // instances a ray starts with (a TLAS of at most PT_ROOT_ITEMS_MAX, see pt_ray_start), or 0: it starts at node 0
    const float4* tris;
    const PtDevInstance* inst;     // in the order of the TLAS leaf slots
};

#define PT_NODE_SENTINEL 0x7FFFFFFF
#define PT_NODE_DONE 0x7FFFFFFE
#define PT_NODE_EMPTY ((int)0x80000000)
#define PT_NO_INSTANCE 0xFFFFFFFFu

// Per-lane traversal stack: the first PT_LDS_STACK entries live in LDS ([entry][thread], one bank
// per lane), deeper entries spill to a per-thread slice of a global buffer.  The traversal kernels
// run with PT_TRAVERSE_BLOCK threads per block, which makes the LDS stride a constant.
#define PT_TRAVERSE_BLOCK 256
typedef __attribute__((address_space(3))) int pt_lds_int;
typedef __attribute__((address_space(3))) float pt_lds_float;

// LDS_N: entries kept in LDS.  SLICES: rays per thread that own a stack (the pair-scheduled traversal keeps
// two rays per lane); `slice` selects the spill slice of the ray in hand.
template <int LDS_N, int SLICES>
struct PtStackT {
    pt_lds_int* lds;   // &stack[0][threadIdx.x] of the ray in hand
    int* spill_base;   // spill buffer of the whole grid; this thread's slice is computed on demand
    int sp;
    int slice;
    __device__ __forceinline__ int* spill_slot(int k) const
    {
        const size_t t = (size_t)(blockIdx.x * PT_TRAVERSE_BLOCK + threadIdx.x);
        return spill_base + (SLICES == 1 ? t : t * SLICES + (size_t)slice) * PT_SPILL_STACK + k;
    }
    __device__ __forceinline__ void push(int v)
    {
        if (sp < LDS_N) lds[sp * PT_TRAVERSE_BLOCK] = v;
        else *spill_slot(sp - LDS_N) = v;
        ++sp;
    }
    // Pushes r[m], ..., r[1] in this order (m in 0..3) so that r[1] is popped first.
    __device__ __forceinline__ void push_far_to_near(int r1, int r2, int r3, int m)
    {
        if (sp + 3 <= LDS_N) {
            // r[j] belongs at sp + m - j; an entry beyond m lands on slot sp first and is
            // overwritten by r[m], or stays above the new top when m == 0
            lds[(sp + max(m - 3, 0)) * PT_TRAVERSE_BLOCK] = r3;
            lds[(sp + max(m - 2, 0)) * PT_TRAVERSE_BLOCK] = r2;
            lds[(sp + max(m - 1, 0)) * PT_TRAVERSE_BLOCK] = r1;
            sp += m;
        } else {
            if (m > 2) push(r3);
            if (m > 1) push(r2);
            if (m > 0) push(r1);
        }
    }
    __device__ __forceinline__ int pop()
    {
        if (sp == 0) return PT_NODE_DONE;
        --sp;
        int v;
        if (sp < LDS_N) v = lds[sp * PT_TRAVERSE_BLOCK];
        else v = *spill_slot(sp - LDS_N);
        return v;
    }
};
typedef PtStackT<PT_LDS_STACK, 1> PtStack;

// One step through a 64-byte 4-wide node with child boxes quantised to 8 bits on the node's own
// grid: (origin.xyz, step.x) (child[0..3]) (lo.x lo.y lo.z hi.x words) (hi.y hi.z words, step.y, step.z);
// byte k of each word belongs to child k and a plane is fma(byte, step, origin).
// Continues with the nearest child that the ray interval [tlo, hi] overlaps and pushes the others
// so that nearer ones are popped first; returns the next item (a pop if nothing was hit).
// What a ray starts with.  Over a handful of instances the box steps of the TLAS only order and cull what the
// instance phase tests again with the exact boxes, at the price of full steps per ray: such a ray starts with the
// instances themselves, in leaf-slot order (the closest hit does not depend on the order).  A TLAS leaf reference
// is ~(slot << 3).
template <typename STACK>
__device__ __forceinline__ int pt_ray_start(const PtTlasView& tl, STACK& stk)
{
    stk.sp = 0;
    if (tl.n_root == 0) return 0;
    for (int k = tl.n_root - 1; k >= 1; --k) stk.push((int)~((uint32_t)k << 3));
    return (int)~0u;
}

// F32 = false: the 64-byte node above.  F32 = true: the same node as 128 bytes with its planes as floats
// (PtWideF: [lo.x hi.x lo.y hi.y lo.z hi.z] x four children, then the references).  The nearer plane vector of an
// axis is the low one when 1/d > 0, and the ray's sign picks it by address, so the step needs neither the 24 byte
// conversions and first fmas nor the word selects: 99 vector instructions instead of 153.  It also loads seven
// 16-byte vectors per lane instead of four, which costs incoherent rays more than the instructions save
// (k_intersect +3 %); where the 64 lanes of a wave walk the same nodes the loads are broadcasts (k_primary -10 %).
template <bool F32, typename STACK>
__device__ __forceinline__ int pt_wide_step(const PtRaySpace& r, const float4* __restrict__ nodes, int node,
                                            float tlo, float hi, STACK& stk)
{
    float tn[4]; int ref[4];
    const float inf = __builtin_inff();
    if constexpr (F32) {
    static_assert(((unsigned long long)PT_MAX_POOL_NODES << 7) <= (1ull << 32), "pool nodes are addressed with 32-bit byte offsets");
    const uint32_t base = (uint32_t)node << 7;
    const uint32_t sx = (__float_as_uint(r.ix) >> 27) & 16u, sy = (__float_as_uint(r.iy) >> 27) & 16u, sz = (__float_as_uint(r.iz) >> 27) & 16u;
    const char* pool = (const char*)nodes;
    // (offsets are unsigned 32-bit: the subtraction comes after the addition, so that node 0 does not wrap)
    const float4 nx4 = *(const float4*)(pool + (base + sx)), fx4 = *(const float4*)(pool + ((base + 16u) - sx));
    const float4 ny4 = *(const float4*)(pool + (base + 32u + sy)), fy4 = *(const float4*)(pool + ((base + 48u) - sy));
    const float4 nz4 = *(const float4*)(pool + (base + 64u + sz)), fz4 = *(const float4*)(pool + ((base + 80u) - sz));
    const float4 rf = *(const float4*)(pool + (base + 96u));
    const float nxv[4] = { nx4.x, nx4.y, nx4.z, nx4.w }, fxv[4] = { fx4.x, fx4.y, fx4.z, fx4.w };
    const float nyv[4] = { ny4.x, ny4.y, ny4.z, ny4.w }, fyv[4] = { fy4.x, fy4.y, fy4.z, fy4.w };
    const float nzv[4] = { nz4.x, nz4.y, nz4.z, nz4.w }, fzv[4] = { fz4.x, fz4.y, fz4.z, fz4.w };
    ref[0] = __float_as_int(rf.x); ref[1] = __float_as_int(rf.y); ref[2] = __float_as_int(rf.z); ref[3] = __float_as_int(rf.w);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float ax = PT_FMA(nxv[k], r.ix, -r.oox), ay = PT_FMA(nyv[k], r.iy, -r.ooy), az = PT_FMA(nzv[k], r.iz, -r.ooz);
        const float bx = PT_FMA(fxv[k], r.ix, -r.oox), by = PT_FMA(fyv[k], r.iy, -r.ooy), bz = PT_FMA(fzv[k], r.iz, -r.ooz);
        const float a = fmaxf(fmaxf(ax, ay), az), b = fminf(fminf(bx, by), bz);
        // An empty child is (+inf, -inf) and needs no test of its own (testing the reference as the 64-byte step
        // does cost k_primary 6 %): a child is followed only if its entry distance compares below +inf, and an
        // empty child's is +inf for a finite ray, NaN or an infinity for a ray that is not finite.
        const bool hit = fmaxf(a, tlo) <= fminf(b, hi);
        tn[k] = hit ? a : inf;
    }
    } else {
    const float4 h0 = nodes[4 * (size_t)node + 0], h1 = nodes[4 * (size_t)node + 1];
    const float4 h2 = nodes[4 * (size_t)node + 2], h3 = nodes[4 * (size_t)node + 3];
    const float sx = h0.w, sy = h3.z, sz = h3.w;
    const uint32_t Lx = __float_as_uint(h2.x), Ly = __float_as_uint(h2.y), Lz = __float_as_uint(h2.z);
    const uint32_t Hx = __float_as_uint(h2.w), Hy = __float_as_uint(h3.x), Hz = __float_as_uint(h3.y);
    const int ref_in[4] = { __float_as_int(h1.x), __float_as_int(h1.y), __float_as_int(h1.z), __float_as_int(h1.w) };
    // The plane distance fma(x, 1/d, -o/d) is monotone in x, so the nearer plane of an axis is the
    // low one when 1/d > 0 and the high one otherwise: picking the plane words by the sign gives
    // the same interval as taking min and max of both distances.
    const bool nx = r.ix < 0.0f, ny = r.iy < 0.0f, nz = r.iz < 0.0f;
    const uint32_t Nx = nx ? Hx : Lx, Fx = nx ? Lx : Hx;
    const uint32_t Ny = ny ? Hy : Ly, Fy = ny ? Ly : Hy;
    const uint32_t Nz = nz ? Hz : Lz, Fz = nz ? Lz : Hz;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float ax = PT_FMA(PT_FMA((float)((Nx >> (8 * k)) & 255u), sx, h0.x), r.ix, -r.oox);
        const float ay = PT_FMA(PT_FMA((float)((Ny >> (8 * k)) & 255u), sy, h0.y), r.iy, -r.ooy);
        const float az = PT_FMA(PT_FMA((float)((Nz >> (8 * k)) & 255u), sz, h0.z), r.iz, -r.ooz);
        const float bx = PT_FMA(PT_FMA((float)((Fx >> (8 * k)) & 255u), sx, h0.x), r.ix, -r.oox);
        const float by = PT_FMA(PT_FMA((float)((Fy >> (8 * k)) & 255u), sy, h0.y), r.iy, -r.ooy);
        const float bz = PT_FMA(PT_FMA((float)((Fz >> (8 * k)) & 255u), sz, h0.z), r.iz, -r.ooz);
        const float a = fmaxf(fmaxf(ax, ay), az), b = fminf(fminf(bx, by), bz);
        const bool hit = fmaxf(a, tlo) <= fminf(b, hi);      // an unused slot is an inverted box in the pool (k_wide_rebase): no test of the reference
        tn[k] = hit ? a : inf;
        ref[k] = ref_in[k];
    }
    }
    const int nh = (tn[0] < inf ? 1 : 0) + (tn[1] < inf ? 1 : 0) + (tn[2] < inf ? 1 : 0) + (tn[3] < inf ? 1 : 0);
#ifndef PT_NO_UNIFORM_SKIP
    // Ordering matters only where a lane has two or more children to go on with; when no lane of the wave has
    // (coherent rays near the leaves: +1.5 % on the dragon frame), the five compare-exchanges are skipped by a
    // wave-uniform branch.
    if (__ballot(nh > 1) == 0ull) {
        const int r = tn[0] < inf ? ref[0] : (tn[1] < inf ? ref[1] : (tn[2] < inf ? ref[2] : ref[3]));
        return nh ? r : stk.pop();
    }
#endif
#define PT_CSWAP(i, j) { const bool sw = tn[j] < tn[i]; const float ta = sw ? tn[j] : tn[i], tb = sw ? tn[i] : tn[j]; \
                         const int ra = sw ? ref[j] : ref[i], rb = sw ? ref[i] : ref[j]; tn[i] = ta; tn[j] = tb; ref[i] = ra; ref[j] = rb; }
    PT_CSWAP(0, 1) PT_CSWAP(2, 3) PT_CSWAP(0, 2) PT_CSWAP(1, 3) PT_CSWAP(1, 2)
#undef PT_CSWAP
    // The children to go on with are the nh smallest keys, at the front: the count taken before the network holds
    // after it.  (For a ray that is not finite some keys can be NaN and the order is then not total; every slot of a
    // pool node holds a valid reference, so whatever such a ray goes on with, it reads no memory it should not.)
    stk.push_far_to_near(ref[1], ref[2], ref[3], max(nh - 1, 0));
    return nh > 0 ? ref[0] : stk.pop();
}

// Moves a world-space ray into the object space of an instance (world->object rows r0..r2).
__device__ __forceinline__ void pt_to_object_space(const float4 r0, const float4 r1, const float4 r2, const PtRaySpace& ws,
                                                   float& ox, float& oy, float& oz, float& dx, float& dy, float& dz)
{
    ox = PT_FMA(r0.x, ws.ox, PT_FMA(r0.y, ws.oy, PT_FMA(r0.z, ws.oz, r0.w)));
    oy = PT_FMA(r1.x, ws.ox, PT_FMA(r1.y, ws.oy, PT_FMA(r1.z, ws.oz, r1.w)));
    oz = PT_FMA(r2.x, ws.ox, PT_FMA(r2.y, ws.oy, PT_FMA(r2.z, ws.oz, r2.w)));
    dx = PT_FMA(r0.x, ws.dx, PT_FMA(r0.y, ws.dy, r0.z * ws.dz));
    dy = PT_FMA(r1.x, ws.dx, PT_FMA(r1.y, ws.dy, r1.z * ws.dz));
    dz = PT_FMA(r2.x, ws.dx, PT_FMA(r2.y, ws.dy, r2.z * ws.dz));
}

// traceRayEXT(topLevelAS, gl_RayFlagsOpaqueEXT, 0xFF, 0, 0, 0, o, tmin, d, tmax, 0) of
// shaders/ray_gen.rgen:54-65 with face culling disabled per instance
// (source/top_level_acceleration_structure.cpp:28).
// Closest hit = lexicographic minimum of (t, instance, primitive) over every triangle passing the
// gated test with tmin < t < tmax and t inside the instance's world-box interval.
// Plain form: one ray per lane from start to end (the wave-scheduled form is in pt_traverse_sched.h).
template <bool F32>
__device__ __forceinline__ void pt_closest_hit(const PtTlasView& tl, PtStack& stk, float ox, float oy, float oz,
                                               float dx, float dy, float dz, float tmin, float tmax, PtHitRec& best)
{
    best.t = tmax; best.inst = PT_NO_INSTANCE; best.prim = 0xFFFFFFFFu; best.u = 0.0f; best.v = 0.0f;
    PtRaySpace ws, cur;
    pt_ray_space_init(ws, ox, oy, oz, dx, dy, dz);
    cur = ws;
    float delta = 0.0f;
    float tlo = tmin, thi = tmax;   // closed interval handed to box tests in the current space
    uint32_t cur_inst = PT_NO_INSTANCE;
    int node = pt_ray_start(tl, stk);
    for (;;) {
        while ((unsigned)node < (unsigned)PT_NODE_DONE)
            node = pt_wide_step<F32>(cur, F32 ? tl.nodes_f32 : tl.nodes, node, tlo, fminf(thi, best.t), stk);
        if (node == PT_NODE_DONE) break;
        if (node == PT_NODE_SENTINEL) {      // all of the instance's BLAS visited: back to world space
            cur_inst = PT_NO_INSTANCE; cur = ws; tlo = tmin; thi = tmax;
            node = stk.pop();
            continue;
        }
        const uint32_t code = ~(uint32_t)node;
        const uint32_t first = code >> 3, cnt = (code & 7u) + 1u;
        if (cur_inst == PT_NO_INSTANCE) {
            const float4* q = tl.inst[first].q;
            const float4 blo = q[3], bhi = q[4];
            const uint32_t ii = __float_as_uint(q[5].y);
            float a, b;
            pt_slab(ws, blo.x, blo.y, blo.z, bhi.x, bhi.y, bhi.z, a, b);
            if ((a <= b) && !(b < tmin) && !(a > tmax) && !(a > best.t)) {
                float oox, ooy, ooz, odx, ody, odz;
                pt_to_object_space(q[0], q[1], q[2], ws, oox, ooy, ooz, odx, ody, odz);
                if (pt_ray_valid(odx, ody, odz)) {
                    pt_ray_space_init(cur, oox, ooy, ooz, odx, ody, odz);
                    delta = blo.w;
                    cur_inst = ii;
                    tlo = fmaxf(tmin, a); thi = fminf(tmax, b);
                    stk.push(PT_NODE_SENTINEL);
                    node = __float_as_int(q[5].x);
                    continue;
                }
            }
            node = stk.pop();
            continue;
        }
        for (uint32_t k = 0; k < cnt; ++k) {
            const float4 a = tl.tris[3 * (size_t)(first + k) + 0], b = tl.tris[3 * (size_t)(first + k) + 1], c = tl.tris[3 * (size_t)(first + k) + 2];
            float t, u, v;
            if (pt_tri_test(cur, delta, a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, t, u, v)) {
                const uint32_t prim = __float_as_uint(c.y);
                // tlo / thi fold the instance's world-box interval into (tmin, tmax)
                if (t > tmin && t < tmax && t >= tlo && t <= thi) {
                    pt_hit_update(best, t, cur_inst, prim, u, v);
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
    float inv = pt_rcp(pt_sqrt(len2));
    x *= inv; y *= inv; z *= inv;
}
