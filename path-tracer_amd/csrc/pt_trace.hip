// pt_trace.hip -- the wavefront path-tracing pipeline that replaces vkCmdTraceRaysKHR
// (source/renderer.cpp:122) and the three shaders it runs:
//   k_primary    bounce 0 in one kernel: shaders/ray_gen.rgen:21-47 (primary ray, per-sample seed, payload
//                reset: pure functions of the path index, computed in registers), the traceRayEXT of
//                ray_gen.rgen:54-65 and the hit / miss shading of that bounce
//   k_intersect  shaders/ray_gen.rgen:54-65   traceRayEXT of every later bounce (software two-level BVH traversal)
//   k_shade      shaders/closest_hit.rchit:55-86 and shaders/miss.rmiss:8-21, plus the bounce
//                bookkeeping of ray_gen.rgen:67-68; survivors are compacted into the next queue
//   k_accumulate shaders/ray_gen.rgen:71-87   per-pixel sum over samples, running mean / sum
//
// Path state lives in four float4 streams per queue slot:
//   q0 = (origin.xyz, -)                         q1 = (direction.xyz, -)            read by traversal, one buffer each
//   q2 = (currentWeight.rgb, bits(seed))         q3 = (hitValue.rgb, bits(path id)) read by shading, ping-pong between bounces
// so every lane moves 16 bytes per access and a wave moves 1 KiB per instruction.  The "path id" is the
// slot of the path's final radiance, laid out [pixel of the chunk][sample].  The streams come into being
// with the survivors of bounce 0; bounce 0 itself reads none of them.
//
// Ray queues are split into kShards independent shards, each with its own size and work counter
// on its own 128-byte line.  A single queue head saturates at roughly 90 atomics per microsecond,
// which for 33 M rays (half a million 64-ray batches) costs more than the traversal itself;
// 64 heads remove that limit.  Traversal runs on a fixed persistent grid: every wave starts at the
// shard matching its index, claims slots with one atomic per few batches and moves on to the next
// shard when one runs dry.  Shading runs on a resident grid whose waves take the batches of their
// shard by index, without atomics.  Either way sizes stay on the device and a frame needs no host
// round trip.  Survivors of shard s go to shard s of the next queue, which therefore never overflows.
#include "pt_device.h"
#include "pt_traverse_sched.h"
#ifdef PT_EXPERIMENTAL
#include "pt_traverse_pair.h"     // experimental/: two rays per lane, measured slower (make EXPERIMENTAL=1)
#include "pt_traverse_pool.h"     // experimental/: rays pooled in LDS, lanes as workers, measured slower
#endif
#include <string.h>
#include <algorithm>
#include <mutex>

namespace {

#ifndef PT_INTERSECT_WAVES
#define PT_INTERSECT_WAVES 5   // waves per SIMD k_intersect is compiled for: the scheduled traversal needs 92 registers; the plain
                               // schedule in the same kernel (short levels only) pays 44 bytes of scratch per lane for it
#endif
constexpr int kBlock = 256;
static_assert(kBlock == PT_TRAVERSE_BLOCK, "traversal stack stride");
constexpr int kShards = 64;                              // one per lane: the traversal kernels scan them with one load per lane
constexpr int kCtrPad = 32;                              // uint32 slots per counter: one 128-byte line each
constexpr int kCtrPerKind = (PT_MAX_DEPTH + 1) * kShards;
enum { CTR_COUNT = 0, CTR_WORK_INTERSECT = 1, CTR_KINDS = 2 };
constexpr size_t kCounterWords = (size_t)CTR_KINDS * kCtrPerKind * kCtrPad;

__host__ __device__ inline size_t ctr_index(int kind, uint32_t depth, uint32_t shard)
{
    return ((size_t)kind * kCtrPerKind + (size_t)depth * kShards + shard) * kCtrPad;
}

// Groups of PT_DEAL_GROUP consecutive batches of 64 paths are dealt round-robin to the shards, so
// neighbouring batches (neighbouring pixels) sit next to each other in a shard and a wave's chunk holds
// rays that start close together; the partial last group is the last of its shard, so every shard is dense.
#ifndef PT_DEAL_GROUP
#define PT_DEAL_GROUP 4u
#endif
constexpr uint32_t kDealPaths = PT_DEAL_GROUP * 64u;
__host__ __device__ inline uint32_t slot_of_path(uint32_t q, uint32_t shard_cap)
{
    const uint32_t g = q / kDealPaths;
    return (g % kShards) * shard_cap + (g / kShards) * kDealPaths + q % kDealPaths;
}
__host__ __device__ inline uint32_t shard_count(uint32_t n, uint32_t s)
{
    const uint32_t full = n / kDealPaths, rem = n % kDealPaths;
    uint32_t c = (full / kShards + (s < full % kShards ? 1u : 0u)) * kDealPaths;
    if (rem && (full % kShards) == s) c += rem;
    return c;
}

// Calls body(slot index, valid, shard) for every 64-slot batch of one queue level.
// FLAT: batches are dealt to the waves of the grid by index (the grid's wave count must be a multiple of
// kShards) instead of being claimed with atomics; for passes whose batches cost about the same.
template <bool FLAT, typename Body>
__device__ __forceinline__ void for_each_batch(uint32_t* __restrict__ counters, int work_kind, uint32_t depth,
                                               uint32_t shard_cap, Body body)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (FLAT) {
        // wave g of the grid serves shard g % kShards and every (grid waves / kShards)-th batch of it,
        // starting at batch g / kShards: no claiming, and no wave is launched only to find nothing.
        // The wave index is uniform; said so, the shard's size comes through the scalar cache.
        const uint32_t uw = __builtin_amdgcn_readfirstlane(wave);
        const uint32_t s = uw % kShards;
        const uint32_t stride = (gridDim.x * (blockDim.x >> 6)) / kShards * 64u;
        const uint32_t n_s = counters[ctr_index(CTR_COUNT, depth, s)];
        for (uint32_t base = (uw / kShards) * 64u; base < n_s; base += stride)
            body(s * shard_cap + base + lane, base + lane < n_s, s);
        return;
    }
    for (uint32_t k = 0; k < kShards; ++k) {
        const uint32_t s = (wave + k) & (kShards - 1);
        const uint32_t n_s = counters[ctr_index(CTR_COUNT, depth, s)];
        if (n_s == 0) continue;
        uint32_t* work = &counters[ctr_index(work_kind, depth, s)];
        for (;;) {
            // a drained shard costs a load, not an atomic
            if (__hip_atomic_load(work, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n_s) break;
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(work, 64u);
            base = __builtin_amdgcn_readfirstlane(base);
            if (base >= n_s) break;
            body(s * shard_cap + base + lane, base + lane < n_s, s);
        }
    }
}

// ------------------------------------------------------------------------------------------------
struct PtCameraArg { float m[32]; };

// What bounce 0 needs to know about a path without any stored state: ray_gen.rgen:21-47 is a pure
// function of the path index, so the primary ray is computed where it is traced and the seed where
// it is first used; no generation pass writes 32 bytes per path for the next two kernels to read back.
// Division of a 31-bit index by a divisor fixed for the launch (samples per pixel, image width): with
// m = floor(2^32 (2^s - d) / d) + 1, s = ceil(log2 d), floor(n / d) = (mulhi(m, n) + n) >> s for every n < 2^31
// (Granlund and Montgomery's round-up multiplier; the sum stays below 2^32).  Three instructions where the
// compiler's expansion of a 32-bit division takes about thirty.
struct PtDivisor { uint32_t d, m, s; };
inline PtDivisor pt_divisor(uint32_t d)
{
    PtDivisor r; r.d = d; r.s = 0;
    while ((1ull << r.s) < d) ++r.s;
    r.m = (uint32_t)((((1ull << r.s) - d) << 32) / d + 1ull);
    return r;
}
__device__ __forceinline__ uint32_t pt_div(uint32_t n, const PtDivisor& D) { return (__umulhi(D.m, n) + n) >> D.s; }

struct PtPrimaryArg {
    PtCameraArg camera;     // viewInverse, projInverse; by value: no copy to wait for
    uint32_t n_paths, pixel0, spp, W, H, frame, seed_mode;
    PtDivisor by_spp, by_W; // path indices and pixel indices are below 2^31
};

// Path index of a queue slot at bounce 0 (inverse of slot_of_path); s = shard of the slot.
__device__ __forceinline__ uint32_t path_of_slot(uint32_t slot, uint32_t s, uint32_t shard_cap)
{
    const uint32_t within = slot - s * shard_cap;
    return ((within / kDealPaths) * kShards + s) * kDealPaths + within % kDealPaths;
}

// ray_gen.rgen:21-31: origin and direction of the primary ray of path q
__device__ __forceinline__ void pt_primary_ray(const PtPrimaryArg& P, uint32_t q, float& ox, float& oy, float& oz, float& rx, float& ry, float& rz)
{
    const uint32_t pixel = P.pixel0 + pt_div(q, P.by_spp);
    const uint32_t py = pt_div(pixel, P.by_W), px = pixel - py * P.W;
    const float* VI = P.camera.m; const float* PI = P.camera.m + 16;
    float uvx = ((float)px + 0.5f) / (float)P.W;
    float uvy = ((float)py + 0.5f) / (float)P.H;
    float dx = uvx * 2.0f - 1.0f, dy = uvy * 2.0f - 1.0f;
    float tx = PT_FMA(PI[0], dx, PT_FMA(PI[4], dy, PI[8] + PI[12]));
    float ty = PT_FMA(PI[1], dx, PT_FMA(PI[5], dy, PI[9] + PI[13]));
    float tz = PT_FMA(PI[2], dx, PT_FMA(PI[6], dy, PI[10] + PI[14]));
    pt_normalize(tx, ty, tz);
    rx = PT_FMA(VI[0], tx, PT_FMA(VI[4], ty, VI[8] * tz));
    ry = PT_FMA(VI[1], tx, PT_FMA(VI[5], ty, VI[9] * tz));
    rz = PT_FMA(VI[2], tx, PT_FMA(VI[6], ty, VI[10] * tz));
    ox = VI[12]; oy = VI[13]; oz = VI[14];
}

// ray_gen.rgen:38-40 (per-sample seed) and the slot of the path's final radiance, [pixel of the chunk][sample]:
// the samples of a pixel sit in neighbouring lanes when their paths end, so they are written together
__device__ __forceinline__ void pt_primary_seed(const PtPrimaryArg& P, uint32_t q, uint32_t& seed, uint32_t& result_slot)
{
    const uint32_t in_chunk = pt_div(q, P.by_spp), s = q - in_chunk * P.spp, pixel = P.pixel0 + in_chunk;
    // the reference reads the clock once per sample inside a pixel's loop, so successive samples see successive
    // values; here a pixel's samples are lanes that read the (wave-uniform) clock together, and the sample index
    // is added so that they still differ
    const uint32_t v1 = P.seed_mode == PT_SEED_CLOCK ? (uint32_t)clock64() + s : P.frame * P.spp + s;
    seed = pt_tea(pixel, v1);                  // gl_LaunchIDEXT.y * width + x is the pixel's index
    result_slot = q;
}

__global__ void k_init_counts(uint32_t n_paths, uint32_t* __restrict__ counters)
{
    if (threadIdx.x < kShards) counters[ctr_index(CTR_COUNT, 0, threadIdx.x)] = shard_count(n_paths, threadIdx.x);
}

// ------------------------------------------------------------------------------------------------
// traceRayEXT for every ray of one bounce.  Two schedules share the kernel and the per-lane arithmetic:
//   plain      a wave takes 64 rays and keeps them to the end.  Best for coherent primary rays and
//              for levels with few rays per wave (deep bounces), where replacement has nothing to
//              draw from and its bookkeeping only costs;
//   scheduled  ray replacement and phase voting (pt_traverse_sched.h).  Best for the large
//              incoherent levels in between (bounce 1-3 of the dragon frame: 25-30 % faster).
// The level's size lives on the device, so the kernel decides: scheduled when the level holds at
// least sched_min_rays rays (0 forces it, UINT32_MAX forces the plain schedule).
// BY_ID: hit records go to hits[path id] (pt_intersect: caller's ray order) instead of the queue slot.
template <bool BY_ID>
__global__ void __launch_bounds__(kBlock, PT_INTERSECT_WAVES)
k_intersect(PtTlasView tl, uint32_t* __restrict__ counters, uint32_t depth, uint32_t shard_cap,
            const float4* __restrict__ q0, const float4* __restrict__ q1, float tmin, float tmax,
            PtHit* __restrict__ hits, float* __restrict__ t_out, int* __restrict__ spill, uint32_t sched_min_rays)
{
    __shared__ int s_stack[PT_LDS_STACK][kBlock];
    __shared__ float s_world[9][kBlock];    // scheduled traversal: world-space origin, direction and reciprocals of a lane inside an instance
    __shared__ int s_cold[6][kBlock];       // ... and its rarely used state (closest hit's ids and barycentrics, queue slot, delta)
    PtStack stk;
    stk.lds = (pt_lds_int*)&s_stack[0][threadIdx.x];
    stk.spill_base = spill;
    uint32_t total = counters[ctr_index(CTR_COUNT, depth, threadIdx.x & (kShards - 1))];   // lane s: shard s
    for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off);
    if (total >= sched_min_rays) {
        PtSchedQueue Q;
        Q.counters = counters; Q.shard_cap = shard_cap; Q.n_shards = kShards; Q.ctr_pad = kCtrPad;
        Q.count_base = ctr_index(CTR_COUNT, depth, 0); Q.work_base = ctr_index(CTR_WORK_INTERSECT, depth, 0);
        pt_traverse_sched<BY_ID>(tl, Q, stk, (pt_lds_float*)&s_world[0][threadIdx.x], (pt_lds_int*)&s_cold[0][threadIdx.x], q0, q1, tmin, tmax, hits, t_out);
        return;
    }
    for_each_batch<false>(counters, CTR_WORK_INTERSECT, depth, shard_cap, [&](uint32_t i, bool valid, uint32_t shard) {
        if (valid) {
            const float4 o = q0[i], d = q1[i];
            PtHitRec h;
            pt_closest_hit<false>(tl, stk, o.x, o.y, o.z, d.x, d.y, d.z, tmin, tmax, h);
            PtHit out; out.instance = h.inst; out.primitive = h.prim; out.u = h.u; out.v = h.v;
            const uint32_t dst = BY_ID ? __float_as_uint(d.w) : i;
            hits[dst] = out;
            if (BY_ID && t_out) t_out[dst] = h.t;
        }
    });
}

#ifdef PT_EXPERIMENTAL
// traceRayEXT for every ray of one bounce with the rays of a wave pooled in LDS (pt_traverse_pool.h); levels below
// sched_min_rays rays take the plain schedule, whose stack uses the start of the same LDS.
template <bool BY_ID>
__global__ void __launch_bounds__(kBlock, PT_POOL_WAVES)
k_intersect_pool(PtTlasView tl, uint32_t* __restrict__ counters, uint32_t depth, uint32_t shard_cap,
                 const float4* __restrict__ q0, const float4* __restrict__ q1, float tmin, float tmax,
                 PtHit* __restrict__ hits, float* __restrict__ t_out, int* __restrict__ spill, uint32_t sched_min_rays)
{
    __shared__ int s_pool[PT_POOL_WORDS][PT_POOL_COLS];
    __shared__ int s_class[kBlock / 64][32];
    static_assert(sizeof(int) * PT_POOL_WORDS * PT_POOL_COLS >= sizeof(int) * PT_LDS_STACK * kBlock, "the plain schedule's stack lives in the pool's memory");
    uint32_t total = counters[ctr_index(CTR_COUNT, depth, threadIdx.x & (kShards - 1))];   // lane s: shard s
    for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off);
    if (total >= sched_min_rays) {
        PtSchedQueue Q;
        Q.counters = counters; Q.shard_cap = shard_cap; Q.n_shards = kShards; Q.ctr_pad = kCtrPad;
        Q.count_base = ctr_index(CTR_COUNT, depth, 0); Q.work_base = ctr_index(CTR_WORK_INTERSECT, depth, 0);
        const uint32_t w = threadIdx.x >> 6;
        const size_t first_slot = ((size_t)blockIdx.x * (kBlock / 64) + w) * PT_POOL_RAYS;
        pt_traverse_pool<BY_ID>(tl, Q, (pt_lds_int*)&s_pool[0][w * PT_POOL_RAYS], (pt_lds_int*)&s_class[w][0], spill + first_slot * PT_SPILL_STACK,
                                q0, q1, tmin, tmax, hits, t_out);
        return;
    }
    PtStack stk;
    stk.lds = (pt_lds_int*)&s_pool[0][0] + threadIdx.x;
    stk.spill_base = spill;
    for_each_batch<false>(counters, CTR_WORK_INTERSECT, depth, shard_cap, [&](uint32_t i, bool valid, uint32_t shard) {
        if (valid) {
            const float4 o = q0[i], d = q1[i];
            PtHitRec h;
            pt_closest_hit<false>(tl, stk, o.x, o.y, o.z, d.x, d.y, d.z, tmin, tmax, h);
            PtHit out; out.instance = h.inst; out.primitive = h.prim; out.u = h.u; out.v = h.v;
            const uint32_t dst = BY_ID ? __float_as_uint(d.w) : i;
            hits[dst] = out;
            if (BY_ID && t_out) t_out[dst] = h.t;
        }
    });
}
#endif

#ifdef PT_EXPERIMENTAL
// traceRayEXT for every ray of one bounce with two rays per lane (pt_traverse_pair.h): four waves per SIMD,
// 40 KB of LDS per block.
#ifndef PT_PAIR_WAVES
#define PT_PAIR_WAVES 4
#endif
template <bool BY_ID>
__global__ void __launch_bounds__(kBlock, PT_PAIR_WAVES)
k_intersect_pair(PtTlasView tl, uint32_t* __restrict__ counters, uint32_t depth, uint32_t shard_cap,
                 const float4* __restrict__ q0, const float4* __restrict__ q1, float tmin, float tmax,
                 PtHit* __restrict__ hits, float* __restrict__ t_out, int* __restrict__ spill)
{
    __shared__ int s_pair[2 * PT_PAIR_WORDS][kBlock];
    PtSchedQueue Q;
    Q.counters = counters; Q.shard_cap = shard_cap; Q.n_shards = kShards; Q.ctr_pad = kCtrPad;
    Q.count_base = ctr_index(CTR_COUNT, depth, 0); Q.work_base = ctr_index(CTR_WORK_INTERSECT, depth, 0);
    pt_traverse_pair<BY_ID>(tl, Q, (pt_lds_int*)&s_pair[0][threadIdx.x], spill, q0, q1, tmin, tmax, hits, t_out);
}
#endif

// rays given as packed float3 arrays (pt_intersect entry point): repack into the float4 streams
__global__ void k_pack_rays(uint32_t n, uint32_t shard_cap, const float* __restrict__ o, const float* __restrict__ d,
                            float4* __restrict__ q0, float4* __restrict__ q1, uint32_t* __restrict__ counters)
{
    uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < kShards) counters[ctr_index(CTR_COUNT, 0, q)] = shard_count(n, q);
    if (q >= n) return;
    const uint32_t i = slot_of_path(q, shard_cap);
    q0[i] = make_float4(o[3 * (size_t)q], o[3 * (size_t)q + 1], o[3 * (size_t)q + 2], 0.0f);
    q1[i] = make_float4(d[3 * (size_t)q], d[3 * (size_t)q + 1], d[3 * (size_t)q + 2], __uint_as_float(q));
}

// ------------------------------------------------------------------------------------------------
struct __attribute__((aligned(4))) PtVertex { float px, py, pz, nx, ny, nz, tu, tv; };

__device__ __forceinline__ const void* ptr_from(float lo, float hi)
{
    return (const void*)(((uint64_t)__float_as_uint(hi) << 32) | __float_as_uint(lo));
}

// texture(textures[i], uv): LINEAR min/mag, REPEAT, single level, R8G8B8A8_UNORM
// (include/resources/gpu_resources.hpp:44-63, source/model_loader.cpp:18).
__device__ __forceinline__ void pt_texture_rgb(const uint32_t* __restrict__ texels, int W, int H, float u, float v,
                                               float& r, float& g, float& b)
{
    float U = PT_FMA(u, (float)W, -0.5f), V = PT_FMA(v, (float)H, -0.5f);
    float fu = floorf(U), fv = floorf(V);
    float a = U - fu, bb = V - fv;
    long long i0 = (long long)fu, j0 = (long long)fv;
    int x0, x1, y0, y1;
    if (((W & (W - 1)) | (H & (H - 1))) == 0) {
        // power-of-two sizes: the non-negative remainder is a mask (a 64-bit division costs a hundred instructions)
        x0 = (int)(i0 & (W - 1)); x1 = (int)((i0 + 1) & (W - 1));
        y0 = (int)(j0 & (H - 1)); y1 = (int)((j0 + 1) & (H - 1));
    } else {
        x0 = (int)(i0 % W); if (x0 < 0) x0 += W;
        x1 = (int)((i0 + 1) % W); if (x1 < 0) x1 += W;
        y0 = (int)(j0 % H); if (y0 < 0) y0 += H;
        y1 = (int)((j0 + 1) % H); if (y1 < 0) y1 += H;
    }
    uint32_t t00 = texels[(size_t)y0 * W + x0], t10 = texels[(size_t)y0 * W + x1];
    uint32_t t01 = texels[(size_t)y1 * W + x0], t11 = texels[(size_t)y1 * W + x1];
    float out[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float c00 = pt_unorm8((t00 >> (8 * ch)) & 255u), c10 = pt_unorm8((t10 >> (8 * ch)) & 255u);
        float c01 = pt_unorm8((t01 >> (8 * ch)) & 255u), c11 = pt_unorm8((t11 >> (8 * ch)) & 255u);
        float r0 = PT_FMA(a, c10 - c00, c00);
        float r1 = PT_FMA(a, c11 - c01, c01);
        out[ch] = PT_FMA(bb, r1 - r0, r0);
    }
    r = out[0]; g = out[1]; b = out[2];
}

// What closest_hit.rchit / miss.rmiss need besides the hit: the streams of the level being shaded
// (q2 weight + seed, q3 radiance + result slot), the next level's queue (o0..o3) and the results.
struct PtShadeArgs {
    const PtDevShadeInstance* shade_inst;
    uint32_t depth, max_depth;
    uint32_t* counters;
    uint32_t shard_cap;
    const float4 *q2, *q3;
    float4 *o0, *o1, *o2, *o3;
    float4* path_result;
    PtPrimaryArg primary;
};

// miss.rmiss:10-20 followed by ray_gen.rgen:67: the miss colour times the path weight joins the radiance
__device__ __forceinline__ void shade_miss(uint32_t depth, const float4 cw, float4& hv)
{
    const float e = depth == 0 ? 1.0f * 0.8f : 0.01f;
    hv.x = PT_FMA(e, cw.x, hv.x); hv.y = PT_FMA(e, cw.y, hv.y); hv.z = PT_FMA(e, cw.z, hv.z);
}

// closest_hit.rchit:55-86 for one hit, followed by ray_gen.rgen:67: the next ray (origin o, direction d), the factor
// w of the path weight, the seed after its two draws, and the emission times the path weight added to hv.
__device__ __forceinline__ void pt_shade_hit(const PtDevShadeInstance* __restrict__ shade_inst, const PtHit h, uint32_t& seed_io,
                                             const float4 cw, float4& hv, float4& no, float4& nd, float& wr_out, float& wg_out, float& wb_out)
{
    // closest_hit.rchit:57-60 (table chain pre-resolved per instance)
    const PtDevShadeInstance& I = shade_inst[h.instance];
    const float4 ptrs = I.ptrs;
    // UnpackGeometry, closest_hit.rchit:33-53: the primitive's three vertices, gathered when the tables were bound
    const float4* rec = (const float4*)ptr_from(ptrs.x, ptrs.y) + 6 * (size_t)h.primitive;
    const float4 a0 = rec[0], a1 = rec[1], b0_ = rec[2], b1_ = rec[3], c0 = rec[4], c1 = rec[5];
    const PtVertex A = { a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w };
    const PtVertex B = { b0_.x, b0_.y, b0_.z, b0_.w, b1_.x, b1_.y, b1_.z, b1_.w };
    const PtVertex C = { c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w };
    const float b0 = 1.0f - h.u - h.v, b1 = h.u, b2 = h.v;
#define PT_MIX3(x, y, z) PT_FMA((z), b2, PT_FMA((y), b1, (x) * b0))
    float px = PT_MIX3(A.px, B.px, C.px), py = PT_MIX3(A.py, B.py, C.py), pz = PT_MIX3(A.pz, B.pz, C.pz);
    float nx = PT_MIX3(A.nx, B.nx, C.nx), ny = PT_MIX3(A.ny, B.ny, C.ny), nz = PT_MIX3(A.nz, B.nz, C.nz);
    float tu = PT_MIX3(A.tu, B.tu, C.tu), tv = PT_MIX3(A.tv, B.tv, C.tv);
#undef PT_MIX3
    pt_normalize(nx, ny, nz);
    // :62-63
    const float4 m0 = I.o2w[0], m1 = I.o2w[1], m2 = I.o2w[2];
    float wx = PT_FMA(m0.x, px, PT_FMA(m0.y, py, PT_FMA(m0.z, pz, m0.w)));
    float wy = PT_FMA(m1.x, px, PT_FMA(m1.y, py, PT_FMA(m1.z, pz, m1.w)));
    float wz = PT_FMA(m2.x, px, PT_FMA(m2.y, py, PT_FMA(m2.z, pz, m2.w)));
    const float4 w0 = I.w2o[0], w1 = I.w2o[1], w2 = I.w2o[2];
    float Nx = PT_FMA(w0.x, nx, PT_FMA(w1.x, ny, w2.x * nz));
    float Ny = PT_FMA(w0.y, nx, PT_FMA(w1.y, ny, w2.y * nz));
    float Nz = PT_FMA(w0.z, nx, PT_FMA(w1.z, ny, w2.z * nz));
    pt_normalize(Nx, Ny, Nz);
    // CreateCoordinateSystem, sampling.glsl:53-65
    float Tx, Ty, Tz;
    if (fabsf(Nx) > fabsf(Ny)) {
        float inv = pt_rcp(pt_sqrt(PT_FMA(Nz, Nz, Nx * Nx)));
        Tx = Nz * inv; Ty = 0.0f; Tz = -Nx * inv;
    } else {
        float inv = pt_rcp(pt_sqrt(PT_FMA(Nz, Nz, Ny * Ny)));
        Tx = 0.0f; Ty = -Nz * inv; Tz = Ny * inv;
    }
    float Bx = PT_FMA(Ny, Tz, -(Nz * Ty));
    float By = PT_FMA(Nz, Tx, -(Nx * Tz));
    float Bz = PT_FMA(Nx, Ty, -(Ny * Tx));
    // SamplingHemisphere, sampling.glsl:38-50
    uint32_t seed = seed_io;
    float r1 = pt_random(seed);
    float r2 = pt_random(seed);
    float sq = pt_sqrt(r1);
    float sn, cs;
    pt_sincos(6.28318548f * r2, sn, cs);
    float lx = cs * sq, ly = sn * sq, lz = pt_sqrt(1.0f - r1);
    float Dx = PT_FMA(lz, Nx, PT_FMA(ly, Bx, lx * Tx));
    float Dy = PT_FMA(lz, Ny, PT_FMA(ly, By, lx * Ty));
    float Dz = PT_FMA(lz, Nz, PT_FMA(ly, Bz, lx * Tz));
    // :71-85
    float cosTheta = PT_FMA(Dz, Nz, PT_FMA(Dy, Ny, Dx * Nx));
    float pdf = pt_div_pi(cosTheta);
    const float4 alb = I.albedo;
    float ar = alb.x, ag = alb.y, ab = alb.z;
    if (__float_as_uint(alb.w)) {
        const float4 tex = I.tex;
        float tr, tg, tbb;
        pt_texture_rgb((const uint32_t*)ptr_from(tex.x, tex.y), (int)__float_as_uint(tex.z), (int)__float_as_uint(tex.w), tu, tv, tr, tg, tbb);
        ar = ar * tr; ag = ag * tg; ab = ab * tbb;
    }
    float wr = pt_div_pi(ar) * cosTheta / pdf, wg = pt_div_pi(ag) * cosTheta / pdf, wb = pt_div_pi(ab) * cosTheta / pdf;
    // ray_gen.rgen:67-68
    const float4 em = I.emissive;
    hv.x = PT_FMA(em.x, cw.x, hv.x);
    hv.y = PT_FMA(em.y, cw.y, hv.y);
    hv.z = PT_FMA(em.z, cw.z, hv.z);
    no = make_float4(wx, wy, wz, 0.0f);
    nd = make_float4(Dx, Dy, Dz, 0.0f);
    wr_out = wr; wg_out = wg; wb_out = wb;
    seed_io = seed;
}

// Shades the 64 slots of one batch (slot i per lane, hit h) and compacts the surviving paths into the
// same shard of the next queue.  Called by every lane of the wave, valid or not.
// FIRST: bounce 0, where the path weight is 1 and the collected radiance 0 and neither is read.
template <bool FIRST>
__device__ __forceinline__ void shade_batch(const PtShadeArgs& S, const PtDevShadeInstance* __restrict__ shade_inst,
                                            float4* __restrict__ o0, float4* __restrict__ o1, float4* __restrict__ o2, float4* __restrict__ o3,
                                            float4* __restrict__ path_result, uint32_t i, bool valid, uint32_t shard, const PtHit h,
                                            const float4 cw, float4 hv)
{
    const uint32_t lane = threadIdx.x & 63;
    // Which paths go on is known from the hit alone, so the batch claims its slots in the next queue
    // first and the atomic's round trip overlaps the gathers below.
    const bool alive = valid && h.instance != 0xFFFFFFFFu && S.depth + 1 < S.max_depth;
    const unsigned long long bal = __ballot(alive);
    const uint32_t cnt = __popcll(bal);
    uint32_t slot0 = 0;
    if (cnt && lane == 0) slot0 = atomicAdd(&S.counters[ctr_index(CTR_COUNT, S.depth + 1, shard)], cnt);
    float4 no, nd, nw, nh;
    if (valid) {
        // cw, hv: path weight and collected radiance; bounces after the first carry the seed and the
        // result slot in their spare words (the incoming ray is not needed here).  At bounce 0 the result slot
        // is the path index and the seed (16 rounds of TEA) is only worked out for a path that hits something:
        // a miss never draws a random number (miss.rmiss:8-21)
        uint32_t seed_in = 0, path;
        if (FIRST) path = path_of_slot(i, shard, S.shard_cap);
        else { seed_in = __float_as_uint(cw.w); path = __float_as_uint(hv.w); }
        if (h.instance == 0xFFFFFFFFu) {
            shade_miss(S.depth, cw, hv);
            path_result[path] = hv;
        } else {
            if (FIRST) { uint32_t slot_unused; pt_primary_seed(S.primary, path, seed_in, slot_unused); }
            float4 po, pd; float wr, wg, wb; uint32_t seed = seed_in;
            pt_shade_hit(shade_inst, h, seed, cw, hv, po, pd, wr, wg, wb);
            if (alive) {
                no = po;
                nd = pd;
                nw = make_float4(cw.x * wr, cw.y * wg, cw.z * wb, __uint_as_float(seed));
                nh = make_float4(hv.x, hv.y, hv.z, __uint_as_float(path));
            } else {
                path_result[path] = hv;
            }
        }
    }
    // wave-level compaction into the same shard of the next queue: one atomic per batch (issued above)
    if (cnt) {
        slot0 = __builtin_amdgcn_readfirstlane(slot0);
        if (alive) {
            const uint32_t slot = shard * S.shard_cap + slot0 + __popcll(bal & ((1ull << lane) - 1ull));
            pt_stream_store(&o0[slot], no); pt_stream_store(&o1[slot], nd); pt_stream_store(&o2[slot], nw); pt_stream_store(&o3[slot], nh);
        }
    }
}

// A resident grid whose waves take batches by index: claiming batches one atomic at a time cost a
// persistent shade pass two thirds of its time, and one wave per batch of a bounce-0-sized grid
// cost 0.12 ms per level in waves launched only to find their shard shorter than that.
#ifndef PT_SHADE_WAVES
#define PT_SHADE_WAVES 4
#endif
#ifndef PT_SHADE_GRID
#define PT_SHADE_GRID 1          // resident grids per launch
#endif
#ifndef PT_SHADE_DENSE
#define PT_SHADE_DENSE 40u        // percent of hits in a group above which it is shaded without sorting
#endif
#ifndef PT_SHADE_GROUP
#define PT_SHADE_GROUP 4         // batches a wave sorts into misses and hits at a time
#endif
// Bounces after the first are mostly misses (69 % at bounce 1 of the dragon frame), and a miss costs a
// few instructions while a hit costs a chain of gathers: shaded slot by slot, the chain runs with a
// third of its lanes.  A wave therefore takes PT_SHADE_GROUP consecutive batches of its shard,
// finishes the misses at once and collects the hits in LDS, then shades the hits 64 at a time.
__global__ void __launch_bounds__(kBlock, PT_SHADE_WAVES)
k_shade(PtShadeArgs A, const PtHit* __restrict__ hits)
{
    constexpr uint32_t kSpan = PT_SHADE_GROUP * 64u;
    __shared__ uint32_t s_list[kBlock / 64][5][kSpan];      // per wave: slot, instance, primitive, u, v of the collected hits
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t uw = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const uint32_t s = uw % kShards;
    const uint32_t stride = (gridDim.x * (blockDim.x >> 6)) / kShards * kSpan;
    const uint32_t n_s = A.counters[ctr_index(CTR_COUNT, A.depth, s)];
    const float4* __restrict__ q2 = A.q2; const float4* __restrict__ q3 = A.q3;
    float4* __restrict__ path_result = A.path_result;
    uint32_t (*list)[kSpan] = s_list[threadIdx.x >> 6];
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (uint32_t base = (uw / kShards) * kSpan; base < n_s; base += stride) {
        float4 hh[PT_SHADE_GROUP];
#pragma unroll
        for (uint32_t b = 0; b < PT_SHADE_GROUP; ++b) {
            const uint32_t at = base + b * 64u + lane;
            hh[b] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (at < n_s) hh[b] = pt_stream_load((const float4*)&hits[s * A.shard_cap + at]);
        }
        // how dense are the hits?  Above PT_SHADE_DENSE percent the batches are shaded as they lie
        // (their hit records are already here), sorting would only add a trip through LDS
        uint32_t n_valid = 0, n_hits = 0;
#pragma unroll
        for (uint32_t b = 0; b < PT_SHADE_GROUP; ++b) {
            const bool valid = base + b * 64u + lane < n_s;
            n_valid += __popcll(__ballot(valid));
            n_hits += __popcll(__ballot(valid && __float_as_uint(hh[b].x) != 0xFFFFFFFFu));
        }
        if (n_hits * 100u > n_valid * PT_SHADE_DENSE) {
#pragma unroll
            for (uint32_t b = 0; b < PT_SHADE_GROUP; ++b) {
                const uint32_t at = base + b * 64u + lane, i = s * A.shard_cap + at;
                if (base + b * 64u >= n_s) break;
                float4 cw = make_float4(0.0f, 0.0f, 0.0f, 0.0f), hv = cw;
                if (at < n_s) { cw = pt_stream_load(&q2[i]); hv = pt_stream_load(&q3[i]); }
                PtHit h; h.instance = __float_as_uint(hh[b].x); h.primitive = __float_as_uint(hh[b].y); h.u = hh[b].z; h.v = hh[b].w;
                shade_batch<false>(A, A.shade_inst, A.o0, A.o1, A.o2, A.o3, path_result, i, at < n_s, s, h, cw, hv);
            }
            continue;
        }
        uint32_t n_list = 0;
#pragma unroll
        for (uint32_t b = 0; b < PT_SHADE_GROUP; ++b) {
            const uint32_t at = base + b * 64u + lane, i = s * A.shard_cap + at;
            const bool valid = at < n_s;
            const bool miss = __float_as_uint(hh[b].x) == 0xFFFFFFFFu;
            if (valid && miss) {
                const float4 cw = pt_stream_load(&q2[i]);
                float4 hv = pt_stream_load(&q3[i]);
                shade_miss(A.depth, cw, hv);
                path_result[__float_as_uint(hv.w)] = hv;
            }
            const unsigned long long bal = __ballot(valid && !miss);
            if (valid && !miss) {
                const uint32_t at_list = n_list + __popcll(bal & lt_mask);
                list[0][at_list] = i; list[1][at_list] = __float_as_uint(hh[b].x); list[2][at_list] = __float_as_uint(hh[b].y);
                list[3][at_list] = __float_as_uint(hh[b].z); list[4][at_list] = __float_as_uint(hh[b].w);
            }
            n_list += __popcll(bal);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t k = 0; k < n_list; k += 64u) {
            const bool valid = k + lane < n_list;
            uint32_t i = 0;
            PtHit h; h.instance = 0xFFFFFFFFu; h.primitive = 0u; h.u = h.v = 0.0f;
            float4 cw = make_float4(0.0f, 0.0f, 0.0f, 0.0f), hv = cw;
            if (valid) {
                i = list[0][k + lane]; h.instance = list[1][k + lane]; h.primitive = list[2][k + lane];
                h.u = __uint_as_float(list[3][k + lane]); h.v = __uint_as_float(list[4][k + lane]);
                cw = pt_stream_load(&q2[i]); hv = pt_stream_load(&q3[i]);
            }
            shade_batch<false>(A, A.shade_inst, A.o0, A.o1, A.o2, A.o3, path_result, i, valid, s, h, cw, hv);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Bounce 0 in one kernel: the primary ray is computed from the path index, traced with the plain
// schedule (primary rays are coherent) and shaded where it ends, so no hit record is written and
// read back, and the shading traffic overlaps the arithmetic of the other waves' traversal.
#ifndef PT_PRIMARY_STATIC
#define PT_PRIMARY_STATIC true     // batches by index (the image interleaves over the waves, so they balance) instead of claimed with atomics
#endif
#ifndef PT_PRIMARY_WAVES
#define PT_PRIMARY_WAVES 4
#endif
#ifndef PT_PRIMARY_F32
#define PT_PRIMARY_F32 true        // the 64 lanes of a wave hold one pixel's samples and walk the same nodes: float planes, loads are broadcasts
#endif
// SHARED: the reference traces one and the same primary ray for every sample of a pixel (ray_gen.rgen:21-27 has no
// jitter, the sample loop :38-47 starts each path with it), so its closest hit can be found once per pixel
// (k_primary_hits) and handed to the pixel's samples, which are shaded as before.  An option of the context, off
// by default: a ray of the rays/s metric is one traceRayEXT of the reference, and with the option on bounce 0
// executes (and counts) one per pixel instead of one per sample.
template <bool SHARED>
__global__ void __launch_bounds__(kBlock, PT_PRIMARY_WAVES)
k_primary(PtTlasView tl, PtShadeArgs A, float tmin, float tmax, int* __restrict__ spill, const PtHit* __restrict__ pixel_hits)
{
    __shared__ int s_stack[SHARED ? 1 : PT_LDS_STACK][kBlock];
    PtStack stk;
    stk.lds = (pt_lds_int*)&s_stack[0][threadIdx.x];
    stk.spill_base = spill;
    for_each_batch<PT_PRIMARY_STATIC>(A.counters, CTR_WORK_INTERSECT, 0, A.shard_cap, [&](uint32_t i, bool valid, uint32_t shard) {
        PtHit h; h.instance = 0xFFFFFFFFu; h.primitive = 0u; h.u = h.v = 0.0f;
        if (valid) {
            const uint32_t q = path_of_slot(i, shard, A.shard_cap);
            if constexpr (SHARED) {
                h = pixel_hits[pt_div(q, A.primary.by_spp)];     // the lanes of a wave hold a pixel's samples: a broadcast
            } else {
                float ox, oy, oz, dx, dy, dz;
                pt_primary_ray(A.primary, q, ox, oy, oz, dx, dy, dz);
                PtHitRec r;
                pt_closest_hit<PT_PRIMARY_F32>(tl, stk, ox, oy, oz, dx, dy, dz, tmin, tmax, r);
                h.instance = r.inst; h.primitive = r.prim; h.u = r.u; h.v = r.v;
            }
        }
        shade_batch<true>(A, A.shade_inst, A.o0, A.o1, A.o2, A.o3, A.path_result, i, valid, shard, h,
                          make_float4(1.0f, 1.0f, 1.0f, 0.0f), make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    });
}

// The closest hit of the primary ray of every pixel of a chunk (the shared-primary option): one ray per lane,
// neighbouring lanes neighbouring pixels, a grid-stride loop over the chunk (the stack slices are per resident thread).
__global__ void __launch_bounds__(kBlock, PT_PRIMARY_WAVES)
k_primary_hits(PtTlasView tl, PtPrimaryArg P, uint32_t n_pixels, float tmin, float tmax, int* __restrict__ spill, PtHit* __restrict__ pixel_hits)
{
    __shared__ int s_stack[PT_LDS_STACK][kBlock];
    PtStack stk;
    stk.lds = (pt_lds_int*)&s_stack[0][threadIdx.x];
    stk.spill_base = spill;
    for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < n_pixels; p += gridDim.x * kBlock) {
        float ox, oy, oz, dx, dy, dz;
        pt_primary_ray(P, p * P.spp, ox, oy, oz, dx, dy, dz);
        PtHitRec r;
        pt_closest_hit<false>(tl, stk, ox, oy, oz, dx, dy, dz, tmin, tmax, r);
        PtHit h; h.instance = r.inst; h.primitive = r.prim; h.u = r.u; h.v = r.v;
        pixel_hits[p] = h;
    }
}

// ------------------------------------------------------------------------------------------------
// ray_gen.rgen:71-87 for the pixels of one chunk
__global__ void __launch_bounds__(kBlock)
k_accumulate(uint32_t n_pixels, uint32_t pixel0, uint32_t spp, uint32_t frame, uint32_t mode,
             const float4* __restrict__ path_result, float* __restrict__ frame_result, void* __restrict__ accum)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pixels) return;
    float r = 0.0f, g = 0.0f, b = 0.0f;
    // a pixel's samples are consecutive: eight 16-byte loads (one 128-byte line) are issued together,
    // so the line is fetched once however many waves share the L1; the sum keeps the sample order
    const float4* __restrict__ mine = path_result + (size_t)p * spp;
    uint32_t s = 0;
    for (; s + 8 <= spp; s += 8) {
        float4 hv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) hv[k] = mine[s + k];
#pragma unroll
        for (int k = 0; k < 8; ++k) { r = PT_FMA(hv[k].x, 2.5f, r); g = PT_FMA(hv[k].y, 2.5f, g); b = PT_FMA(hv[k].z, 2.5f, b); }
    }
    for (; s < spp; ++s) {
        const float4 hv = mine[s];
        r = PT_FMA(hv.x, 2.5f, r); g = PT_FMA(hv.y, 2.5f, g); b = PT_FMA(hv.z, 2.5f, b);
    }
    const float fs = (float)spp;
    r = r / fs; g = g / fs; b = b / fs;
    const size_t pix = (size_t)pixel0 + p;
    frame_result[3 * pix + 0] = r; frame_result[3 * pix + 1] = g; frame_result[3 * pix + 2] = b;
    if (mode == PT_ACCUM_SUM_F32) {
        float4* a = (float4*)accum;
        float4 v = a[pix];
        v.x += r; v.y += g; v.z += b; v.w += 1.0f;
        a[pix] = v;
    } else if (mode == PT_ACCUM_MEAN_F32) {
        float4* a = (float4*)accum;
        float4 v = a[pix];
        const float w = 1.0f / (float)(frame + 1);
        if (frame > 0) { v.x = PT_FMA(r, w, v.x * (1.0f - w)); v.y = PT_FMA(g, w, v.y * (1.0f - w)); v.z = PT_FMA(b, w, v.z * (1.0f - w)); }
        else { v.x = r; v.y = g; v.z = b; }
        v.w = 1.0f;
        a[pix] = v;
    } else {
        uchar4* a = (uchar4*)accum;
        uchar4 v = a[pix];
        const float w = 1.0f / (float)(frame + 1);
        float c[3] = { r, g, b };
        float old[3] = { (float)v.x / 255.0f, (float)v.y / 255.0f, (float)v.z / 255.0f };
        unsigned char q[3];
        for (int k = 0; k < 3; ++k) {
            float x = frame > 0 ? PT_FMA(c[k], w, old[k] * (1.0f - w)) : c[k];
            x = x > 0.0f ? (x < 1.0f ? x : 1.0f) : 0.0f;
            q[k] = (unsigned char)(int)rintf(x * 255.0f);
        }
        a[pix] = make_uchar4(q[0], q[1], q[2], 255);
    }
}

// primary_rays: rays traced at bounce 0 (one per path, or one per pixel with the shared-primary option)
__global__ void k_stats(const uint32_t* __restrict__ counters, uint32_t max_depth, uint32_t n_paths, uint32_t primary_rays, uint64_t* __restrict__ stats)
{
    const uint32_t d = blockIdx.x, s = threadIdx.x;   // one wave per depth, one lane per shard
    uint32_t c = (d < max_depth && s < kShards) ? counters[ctr_index(CTR_COUNT, d, s)] : 0u;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if (d == 0) c = primary_rays;
    // lanes run this concurrently
    if (s == 0 && d < max_depth) atomicAdd((unsigned long long*)&stats[d], (unsigned long long)c);
    if (s == 0 && d == 0) atomicAdd((unsigned long long*)&stats[PT_MAX_DEPTH], (unsigned long long)n_paths);
}

__global__ void k_resolve(uint32_t n, const float4* __restrict__ sum, uchar4* __restrict__ out, int bgra)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 s = sum[i];
    float c[3] = { s.x / s.w, s.y / s.w, s.z / s.w };
    unsigned char q[3];
    for (int k = 0; k < 3; ++k) {
        float x = c[k] > 0.0f ? (c[k] < 1.0f ? c[k] : 1.0f) : 0.0f;
        q[k] = (unsigned char)(int)rintf(x * 255.0f);
    }
    out[i] = bgra ? make_uchar4(q[2], q[1], q[0], 255) : make_uchar4(q[0], q[1], q[2], 255);
}

// ------------------------------------------------------------------------------------------------
int persistent_blocks(PtContext* ctx, const void* kernel)
{
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    if (per_cu > 8) per_cu = 8;
    return ctx->num_cus * per_cu;
}

#ifdef PT_EXPERIMENTAL
// Grid of the pooled traversal: every slot of every wave owns a slice of the stack-overflow buffer, which was
// sized for one slice per thread of the widest persistent grid.
int pool_blocks(PtContext* ctx, const void* kernel)
{
    const int by_spill = (int)(((uint64_t)ctx->num_cus * 8 * kBlock) / PT_POOL_COLS);
    return std::max(1, std::min(persistent_blocks(ctx, kernel), by_spill));
}
#endif

// rays_only: the two ray streams alone (pt_intersect writes its results to the caller's arrays).
int ensure_workspace(PtContext* ctx, PtLane& l, uint64_t paths, uint64_t pixels, bool rays_only = false)
{
    PtWorkspace& w = ctx->ws;
    if (!w.stats) {
        PT_HIP_CHECK(hipMalloc((void**)&w.stats, sizeof(uint64_t) * (PT_MAX_DEPTH + 1)));
        PT_HIP_CHECK(hipMemset(w.stats, 0, sizeof(uint64_t) * (PT_MAX_DEPTH + 1)));
        w.spill_threads = (uint64_t)ctx->num_cus * 8 * kBlock;
    }
    if (!l.counters) {
        PT_HIP_CHECK(hipMalloc((void**)&l.counters, sizeof(uint32_t) * kCounterWords));
        PT_HIP_CHECK(hipMalloc((void**)&l.spill, sizeof(int) * PT_SPILL_STACK * w.spill_threads));
    }
    // per-shard capacity: the batches dealt to one shard, in slots
    const uint64_t groups = (paths + kDealPaths - 1) / kDealPaths;
    uint64_t cap = ((groups + kShards - 1) / kShards) * kDealPaths;
    if (cap > l.shard_cap || (!rays_only && l.rays_only)) {
        PT_HIP_CHECK(hipStreamSynchronize(l.stream));   // earlier work of this lane still uses the old queues
        for (int a = 0; a < 2; ++a) for (int k = 0; k < 4; ++k) { (void)hipFree(l.q[a][k]); l.q[a][k] = nullptr; }
        (void)hipFree(l.hits); (void)hipFree(l.path_result); l.hits = nullptr; l.path_result = nullptr;
        l.paths_cap = 0; l.shard_cap = 0;
        if (cap < l.shard_cap_wanted) cap = l.shard_cap_wanted;      // a lane never shrinks
        l.shard_cap_wanted = cap;
        const uint64_t slots = cap * kShards;
        l.rays_only = rays_only;
        // The ray streams (0, 1) exist once: traversal has read a level's rays before shading writes the next level's
        // over them.  The shading streams (2, 3) are read and written by the same pass and ping-pong.
        for (int a = 0; a < (rays_only ? 1 : 2); ++a) for (int k = (a == 1 ? 2 : 0); k < (rays_only ? 2 : 4); ++k) PT_HIP_CHECK(hipMalloc((void**)&l.q[a][k], sizeof(float4) * slots));
        if (!rays_only) {
            PT_HIP_CHECK(hipMalloc((void**)&l.hits, sizeof(PtHit) * slots));
            PT_HIP_CHECK(hipMalloc((void**)&l.path_result, sizeof(float4) * slots));
        }
        l.paths_cap = slots; l.shard_cap = (uint32_t)cap;
    }
    if (pixels > w.frame_pixels_cap) {
        int rc = pt_lanes_sync(ctx);                    // frames in flight write the old image
        if (rc) return rc;
        (void)hipFree(w.frame_result); w.frame_result = nullptr; w.frame_pixels_cap = 0;
        PT_HIP_CHECK(hipMalloc((void**)&w.frame_result, sizeof(float) * 3 * pixels));
        w.frame_pixels_cap = pixels;
    }
    return PT_OK;
}

// UnpackGeometry's gathers (closest_hit.rchit:33-53), done once per primitive instead of once per hit:
// record p = the three 32-byte vertices indices[3p..3p+2] select, so shading reads 96 contiguous
// bytes with no index load in front of them.
__global__ void k_tri_records(const uint8_t* __restrict__ vertices, const uint32_t* __restrict__ indices, uint32_t n_prims,
                              float* __restrict__ out)
{
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // (primitive, corner, word)
    if (g >= (size_t)n_prims * 24u) return;
    const size_t corner = g >> 3; const uint32_t word = (uint32_t)g & 7u;
    const float* v = (const float*)(vertices + sizeof(PtVertex) * (size_t)indices[corner]);
    out[g] = v[word];
}

// Resolves blasInstances[custom] -> geometryNodes -> materials -> textures once per instance.
int ensure_shade_records(PtContext* ctx, const PtTlas* tlas_c)
{
    PtTlas* t = const_cast<PtTlas*>(tlas_c);
    if (t->shade_version == ctx->tables_version && ctx->tables_version != 0) return PT_OK;
    std::vector<PtDevShadeInstance> rec(t->n);
    std::vector<PtTlas::TriRecords> fresh;
    auto free_fresh = [&]() { (void)hipStreamSynchronize(ctx->stream); for (auto& f : fresh) (void)hipFree(f.dev); };
    auto split = [](uint64_t p, float& lo, float& hi) { uint32_t a = (uint32_t)p, b = (uint32_t)(p >> 32); memcpy(&lo, &a, 4); memcpy(&hi, &b, 4); };
    auto bits = [](uint32_t v) { float f; memcpy(&f, &v, 4); return f; };
    for (uint32_t i = 0; i < t->n; ++i) {
        const uint32_t custom = t->host_custom[i];
        if (custom >= ctx->h_blas_instances.size()) { pt_set_error("pt_trace: instance %u has custom index %u but only %zu BLASInstance records exist", i, custom, ctx->h_blas_instances.size()); free_fresh(); return PT_ERROR_INVALID_ARGUMENT; }
        const PtGeometryNode& gn = ctx->h_geometry_nodes[ctx->h_blas_instances[custom].firstGeometryIndex];
        const PtMaterial& m = ctx->h_materials[gn.materialIndex];
        PtDevShadeInstance& r = rec[i];
        const float* o = &t->host_o2w[12 * (size_t)i]; const float* w = &t->host_w2o[12 * (size_t)i];
        for (int k = 0; k < 3; ++k) {
            r.o2w[k] = make_float4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
            r.w2o[k] = make_float4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
        }
        // the instance's triangle records: shared with every instance that names the same buffers
        const uint32_t prims = t->host_prim_count[i];
        size_t k = 0;
        while (k < fresh.size() && !(fresh[k].vertices == gn.vertexBufferDeviceAddress && fresh[k].indices == gn.indexBufferDeviceAddress && fresh[k].prims == prims)) ++k;
        if (k == fresh.size()) {
            PtTlas::TriRecords tr; tr.vertices = gn.vertexBufferDeviceAddress; tr.indices = gn.indexBufferDeviceAddress; tr.prims = prims; tr.dev = nullptr;
            if (!tr.vertices || !tr.indices) { pt_set_error("pt_trace: geometry node of instance %u has a null buffer address", i); free_fresh(); return PT_ERROR_INVALID_ARGUMENT; }
            hipError_t e = hipMalloc((void**)&tr.dev, sizeof(float) * 24 * (size_t)prims);
            if (e != hipSuccess) { pt_set_error("pt_trace: %s", hipGetErrorString(e)); free_fresh(); return PT_ERROR_HIP; }
            fresh.push_back(tr);
            k_tri_records<<<(uint32_t)(((size_t)prims * 24u + kBlock - 1) / kBlock), kBlock, 0, ctx->stream>>>((const uint8_t*)(uintptr_t)tr.vertices, (const uint32_t*)(uintptr_t)tr.indices, prims, (float*)tr.dev);
        }
        split((uint64_t)(uintptr_t)fresh[k].dev, r.ptrs.x, r.ptrs.y);
        r.ptrs.z = r.ptrs.w = 0.0f;
        r.albedo = make_float4(m.albedoFactor[0], m.albedoFactor[1], m.albedoFactor[2], bits(m.useAlbedoMap ? 1u : 0u));
        r.emissive = make_float4(m.emissiveFactor[0], m.emissiveFactor[1], m.emissiveFactor[2], 0.0f);
        r.tex = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (m.useAlbedoMap) {
            const PtTextureDesc& td = ctx->h_textures[m.albedoMapIndex];
            split(td.rgba8_dev, r.tex.x, r.tex.y);
            r.tex.z = bits(td.width); r.tex.w = bits(td.height);
        }
    }
    int rc = pt_lanes_sync(ctx);
    if (rc) { free_fresh(); return rc; }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess) { free_fresh(); pt_set_error("pt_trace: building the triangle records failed"); return PT_ERROR_HIP; }
    for (auto& old_rec : t->tri_records) (void)hipFree(old_rec.dev);
    t->tri_records.swap(fresh);
    PT_HIP_CHECK(hipMemcpy(t->shade_inst, rec.data(), sizeof(PtDevShadeInstance) * t->n, hipMemcpyHostToDevice));
    t->shade_version = ctx->tables_version;
    return PT_OK;
}

// Smallest level (in rays) the scheduled traversal is used for at bounce `depth`.
uint32_t sched_threshold(const PtContext* ctx, uint32_t depth)
{
    if (ctx->traversal_mode == 0) return 0xFFFFFFFFu;        // plain everywhere
    if (ctx->traversal_mode == 1) return 0u;                 // scheduled everywhere
    if (ctx->traversal_mode == 4) return 0u;                                              // pooled everywhere
    if (ctx->traversal_mode == 5) return depth == 0 ? 0xFFFFFFFFu : ctx->pool_min_rays;   // pooled above the threshold, plain below
    if (ctx->traversal_mode == 3) return ctx->sched_min_rays; // (experimental two-rays-per-lane kernel: has its own launch; this is for entry points that do not use it)
    return depth == 0 ? 0xFFFFFFFFu : ctx->sched_min_rays;   // primary rays are coherent
}

PtTlasView view_of(const PtTlas* t)
{
    PtTlasView v;
    v.nodes = (const float4*)t->pool_nodes; v.nodes_f32 = (const float4*)t->pool_nodes_f32; v.tris = t->pool_tris; v.inst = t->inst;
    v.n_root = t->n_root;
    return v;
}

} // namespace

// ---- pt_intersect for callers that cannot allocate --------------------------------------------
// The two ray streams, the queue counters and the stack-overflow slices that pt_intersect takes from
// hipMalloc live in one file-scope device array instead; each GPU the library runs on holds its own
// copy until the process ends.  Sized for kNaMaxRays rays and a grid of kNaBlocks blocks:
// rays 2 x 16 MiB, counters 1 MiB, stack slices 56 MiB.  Every call on a GPU uses that GPU's one
// array, so calls that overlap on one GPU (two streams, two contexts, two threads) race.
constexpr uint32_t kNaMaxRays = 1u << 20;
constexpr int kNaBlocks = 512;
constexpr size_t kNaStream = sizeof(float4) * (size_t)kNaMaxRays;                       // one ray stream: shard capacity x kShards slots
constexpr size_t kNaCounters = (sizeof(uint32_t) * kCounterWords + 255) / 256 * 256;
constexpr size_t kNaSpill = sizeof(int) * PT_SPILL_STACK * (size_t)kNaBlocks * kBlock;
constexpr size_t kNaBytes = 2 * kNaStream + kNaCounters + kNaSpill;
static __device__ __align__(256) unsigned char scratch[kNaBytes];

constexpr int kNaMaxDevices = 64;
static unsigned char* g_scratch_ptr[kNaMaxDevices];    // device address of `scratch` on each GPU (the array's name is only a host-side stand-in)
static int g_na_blocks[kNaMaxDevices];
static std::mutex g_na_mutex;                          // contexts may be created from several threads: both tables are filled under it

// Called by pt_ctx_create with the context's device current: looks up this GPU's copy of the array
// once and the grid of the traversal kernel, so that the entry point itself asks the runtime for nothing.
int pt_noalloc_setup(PtContext* ctx)
{
    PT_REQUIRE(ctx->device >= 0 && ctx->device < kNaMaxDevices, "device index beyond the table of scratch addresses");
    std::lock_guard<std::mutex> lock(g_na_mutex);
    if (!g_scratch_ptr[ctx->device]) {
        unsigned char* ptr = nullptr;
        PT_HIP_CHECK(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));
        PT_REQUIRE(ptr != nullptr, "hipGetSymbolAddress returned a null address");
        g_na_blocks[ctx->device] = std::min(kNaBlocks, persistent_blocks(ctx, (const void*)k_intersect<true>));
        g_scratch_ptr[ctx->device] = ptr;
    }
    // the context keeps its own copy: the entry point reads no table another thread may be filling
    ctx->na_scratch = g_scratch_ptr[ctx->device]; ctx->na_blocks = g_na_blocks[ctx->device];
    return PT_OK;
}

extern "C" int pt_intersect_noalloc(PtContext* ctx, const PtTlas* tlas, uint32_t n, const float* origins_dev, const float* dirs_dev,
                                    float tmin, float tmax, PtHit* hits_dev, float* t_dev)
{
    PT_REQUIRE(ctx && tlas && origins_dev && dirs_dev && hits_dev, "null argument");
    PT_REQUIRE(tlas->ctx == ctx, "the TLAS belongs to another context");
    if (n > kNaMaxRays) {
        pt_set_error("pt_intersect_noalloc: %u rays exceed the fixed workspace of %u rays; split the call or use pt_intersect", n, kNaMaxRays);
        return PT_ERROR_LIMIT;
    }
    if (n == 0) return PT_OK;
    unsigned char* base = ctx->na_scratch;
    PT_REQUIRE(base != nullptr && ctx->na_blocks > 0, "the fixed workspace was not set up for this device");
    PT_HIP_CHECK(hipSetDevice(ctx->device));
    float4* q0 = (float4*)base;
    float4* q1 = (float4*)(base + kNaStream);
    uint32_t* counters = (uint32_t*)(base + 2 * kNaStream);
    int* spill = (int*)(base + 2 * kNaStream + kNaCounters);
    // per-shard capacity for n rays, as ensure_workspace sizes it: never more than kNaMaxRays / kShards
    const uint64_t groups = ((uint64_t)n + kDealPaths - 1) / kDealPaths;
    const uint32_t cap = (uint32_t)(((groups + kShards - 1) / kShards) * kDealPaths);
    hipStream_t st = ctx->stream;
    // the array keeps what the last call left: sizes and work counters start at zero
    PT_HIP_CHECK(hipMemsetAsync(counters, 0, sizeof(uint32_t) * kCounterWords, st));
    k_pack_rays<<<(n + kBlock - 1) / kBlock, kBlock, 0, st>>>(n, cap, origins_dev, dirs_dev, q0, q1, counters);
    PtTlasView tv = view_of(tlas);
    const uint32_t thr = ctx->traversal_mode == 0 ? 0xFFFFFFFFu : (ctx->traversal_mode == 1 ? 0u : ctx->sched_min_rays);   // mode 3 (two rays per lane) needs twice the stack slices: not used here
    k_intersect<true><<<ctx->na_blocks, kBlock, 0, st>>>(tv, counters, 0, cap, q0, q1, tmin, tmax, hits_dev, t_dev, spill, thr);
    PT_HIP_CHECK(hipGetLastError());
    return PT_OK;
}

extern "C" {

int pt_intersect(PtContext* ctx, const PtTlas* tlas, uint32_t n, const float* origins_dev, const float* dirs_dev,
                 float tmin, float tmax, PtHit* hits_dev, float* t_dev)
{
    PT_REQUIRE(ctx && tlas && origins_dev && dirs_dev && hits_dev, "null argument");
    PT_REQUIRE(tlas->ctx == ctx, "the TLAS belongs to another context");
    PT_REQUIRE(n < (1u << 31), "at most 2^31 - 1 rays per call (queue slots are 32-bit)");
    if (n == 0) return PT_OK;
    PT_HIP_CHECK(hipSetDevice(ctx->device));
    PtLane* lane = nullptr;
    int rc = pt_lane_begin(ctx, &lane);
    if (rc) return rc;
    PtLane& w = *lane;
    if ((rc = ensure_workspace(ctx, w, n, 0, true))) return rc;
    hipStream_t st = w.stream;
    PT_HIP_CHECK(hipMemsetAsync(w.counters, 0, sizeof(uint32_t) * kCounterWords, st));
    k_pack_rays<<<(n + kBlock - 1) / kBlock, kBlock, 0, st>>>(n, w.shard_cap, origins_dev, dirs_dev, w.q[0][0], w.q[0][1], w.counters);
    {
        // caller-supplied rays are treated as incoherent: scheduled above the threshold
#ifdef PT_EXPERIMENTAL
        if (ctx->traversal_mode == 3) {
            const int blocks = std::min(persistent_blocks(ctx, (const void*)k_intersect_pair<true>), ctx->num_cus * 4);
            k_intersect_pair<true><<<blocks, kBlock, 0, st>>>(view_of(tlas), w.counters, 0, w.shard_cap, w.q[0][0], w.q[0][1],
                                                              tmin, tmax, hits_dev, t_dev, w.spill);
        } else
#endif
#ifdef PT_EXPERIMENTAL
        if (ctx->traversal_mode >= 4) {
            const int blocks = pool_blocks(ctx, (const void*)k_intersect_pool<true>);
            k_intersect_pool<true><<<blocks, kBlock, 0, st>>>(view_of(tlas), w.counters, 0, w.shard_cap, w.q[0][0], w.q[0][1],
                                                              tmin, tmax, hits_dev, t_dev, w.spill, sched_threshold(ctx, 1));
        } else
#endif
        {
            const int blocks = persistent_blocks(ctx, (const void*)k_intersect<true>);
            k_intersect<true><<<blocks, kBlock, 0, st>>>(view_of(tlas), w.counters, 0, w.shard_cap, w.q[0][0], w.q[0][1],
                                                         tmin, tmax, hits_dev, t_dev, w.spill, sched_threshold(ctx, 1));
        }
    }
    PT_HIP_CHECK(hipGetLastError());
    return pt_lanes_join(ctx);   // the hit records are ordered on the caller's stream
}

int pt_trace(PtContext* ctx, const PtTlas* tlas, const float camera[32], uint32_t width, uint32_t height,
             uint32_t frame, uint32_t spp, uint32_t max_depth, uint32_t seed_mode, uint32_t accum_mode, void* accum_dev)
{
    PT_REQUIRE(ctx && tlas && camera && accum_dev, "null argument");
    PT_REQUIRE(tlas->ctx == ctx, "the TLAS belongs to another context");
    PT_REQUIRE(width > 0 && height > 0 && (uint64_t)width * height < (1ull << 31), "bad image size");
    PT_REQUIRE(spp > 0 && spp <= 65536, "samples_per_frame must be in [1, 65536]");
    PT_REQUIRE(max_depth > 0 && max_depth < PT_MAX_DEPTH, "max_depth must be in [1, 63]");
    PT_REQUIRE(seed_mode <= PT_SEED_CLOCK && accum_mode <= PT_ACCUM_MEAN_UNORM8, "bad mode");
    PT_REQUIRE(ctx->tables_version != 0, "pt_scene_set_tables has not been called");
    PT_HIP_CHECK(hipSetDevice(ctx->device));
    int rc = ensure_shade_records(ctx, tlas);
    if (rc) return rc;
    const uint64_t n_pixels = (uint64_t)width * height;
    uint64_t chunk_pixels = ctx->max_paths / spp;
    if (chunk_pixels < 1) chunk_pixels = 1;
    if (chunk_pixels > n_pixels) chunk_pixels = n_pixels;
    if (chunk_pixels * spp >= (1ull << 31)) chunk_pixels = ((1ull << 31) - 1) / spp;   // slot indices are 32-bit
    const PtTlasView tv = view_of(tlas);
    PtPrimaryArg primary;
    memcpy(primary.camera.m, camera, sizeof primary.camera.m);
    primary.spp = spp; primary.W = width; primary.H = height; primary.frame = frame; primary.seed_mode = seed_mode;
    primary.by_spp = pt_divisor(spp); primary.by_W = pt_divisor(width);
    const int blocks_i = persistent_blocks(ctx, (const void*)k_intersect<false>);
#ifdef PT_EXPERIMENTAL
    const int blocks_pool = pool_blocks(ctx, (const void*)k_intersect_pool<false>);
#endif
#ifdef PT_EXPERIMENTAL
    const int blocks_pair = std::min(persistent_blocks(ctx, (const void*)k_intersect_pair<false>), ctx->num_cus * 4);   // two stack slices per thread
#endif
    // whole multiples of kShards waves (four waves per block): the passes that deal batches by index need them
    const bool shared = ctx->share_primary;
    const int blocks_p = std::max(16, persistent_blocks(ctx, shared ? (const void*)k_primary<true> : (const void*)k_primary<false>) / 16 * 16);
    const int blocks_ph = persistent_blocks(ctx, (const void*)k_primary_hits);
    const int blocks_s = std::max(16, persistent_blocks(ctx, (const void*)k_shade) * PT_SHADE_GRID / 16 * 16);
    // Every chunk of the frame goes to the next lane and is left in flight there; only the
    // read-modify-write of the image is chained, in submission order.
    for (uint64_t p0 = 0; p0 < n_pixels; p0 += chunk_pixels) {
        const uint32_t np = (uint32_t)((n_pixels - p0) < chunk_pixels ? (n_pixels - p0) : chunk_pixels);
        const uint32_t n_paths = np * spp;
        PtLane* lane = nullptr;
        if ((rc = pt_lane_begin(ctx, &lane))) return rc;
        PtLane& w = *lane;
        if ((rc = ensure_workspace(ctx, w, (uint64_t)n_paths, n_pixels))) return rc;
        hipStream_t st = w.stream;
        PT_HIP_CHECK(hipMemsetAsync(w.counters, 0, sizeof(uint32_t) * kCounterWords, st));
        primary.n_paths = n_paths; primary.pixel0 = (uint32_t)p0;
        k_init_counts<<<1, 64, 0, st>>>(n_paths, w.counters);
        for (uint32_t d = 0; d < max_depth; ++d) {
            const int a = d & 1, b = a ^ 1;
            hipEvent_t ev_stop = nullptr;
            if (ctx->profiling) {
                if (ctx->ev_used + 2 > ctx->ev_pool.size()) {
                    hipEvent_t e0, e1;
                    PT_HIP_CHECK(hipEventCreate(&e0)); PT_HIP_CHECK(hipEventCreate(&e1));
                    ctx->ev_pool.push_back(e0); ctx->ev_pool.push_back(e1);
                }
                PT_HIP_CHECK(hipEventRecord(ctx->ev_pool[ctx->ev_used], st));
                ev_stop = ctx->ev_pool[ctx->ev_used + 1];
                if (ctx->ev_primary.size() < ctx->ev_used / 2 + 1) ctx->ev_primary.resize(ctx->ev_used / 2 + 1);
                ctx->ev_primary[ctx->ev_used / 2] = d == 0;
                ctx->ev_used += 2;
            }
            PtShadeArgs sa;
            sa.shade_inst = tlas->shade_inst; sa.depth = d; sa.max_depth = max_depth; sa.counters = w.counters; sa.shard_cap = w.shard_cap;
            sa.q2 = w.q[a][2]; sa.q3 = w.q[a][3];
            sa.o0 = w.q[0][0]; sa.o1 = w.q[0][1]; sa.o2 = w.q[b][2]; sa.o3 = w.q[b][3];
            sa.path_result = w.path_result; sa.primary = primary;
            if (d == 0) {
                // the timed interval of bounce 0 holds its shading as well
                if (shared) {
                    // the chunk's pixel hits sit at the start of the hit records, which no kernel touches before bounce 1
                    k_primary_hits<<<std::min(blocks_ph, (int)((np + kBlock - 1) / kBlock)), kBlock, 0, st>>>(tv, primary, np, 0.001f, 10000.0f, w.spill, w.hits);
                    k_primary<true><<<blocks_p, kBlock, 0, st>>>(tv, sa, 0.001f, 10000.0f, w.spill, w.hits);
                } else {
                    k_primary<false><<<blocks_p, kBlock, 0, st>>>(tv, sa, 0.001f, 10000.0f, w.spill, nullptr);
                }
                if (ev_stop) PT_HIP_CHECK(hipEventRecord(ev_stop, st));
            } else {
#ifdef PT_EXPERIMENTAL
                if (ctx->traversal_mode == 3)
                    k_intersect_pair<false><<<blocks_pair, kBlock, 0, st>>>(tv, w.counters, d, w.shard_cap, w.q[0][0], w.q[0][1],
                                                                            0.001f, 10000.0f, w.hits, nullptr, w.spill);
                else
#endif
#ifdef PT_EXPERIMENTAL
                if (ctx->traversal_mode >= 4)
                    k_intersect_pool<false><<<blocks_pool, kBlock, 0, st>>>(tv, w.counters, d, w.shard_cap, w.q[0][0], w.q[0][1],
                                                                            0.001f, 10000.0f, w.hits, nullptr, w.spill, sched_threshold(ctx, d));
                else
#endif
                    k_intersect<false><<<blocks_i, kBlock, 0, st>>>(tv, w.counters, d, w.shard_cap, w.q[0][0], w.q[0][1],
                                                                    0.001f, 10000.0f, w.hits, nullptr, w.spill, sched_threshold(ctx, d));
                if (ev_stop) PT_HIP_CHECK(hipEventRecord(ev_stop, st));
                k_shade<<<blocks_s, kBlock, 0, st>>>(sa, w.hits);
            }
        }
        if (ctx->last_accumulate && ctx->last_accumulate != lane)
            PT_HIP_CHECK(hipStreamWaitEvent(st, ctx->last_accumulate->accumulated, 0));
        k_accumulate<<<(np + kBlock - 1) / kBlock, kBlock, 0, st>>>(np, (uint32_t)p0, spp, frame, accum_mode, w.path_result, ctx->ws.frame_result, accum_dev);
        PT_HIP_CHECK(hipEventRecord(w.accumulated, st));
        ctx->last_accumulate = lane;
        k_stats<<<max_depth, 64, 0, st>>>(w.counters, max_depth, n_paths, shared ? np : n_paths, ctx->ws.stats);
    }
    PT_HIP_CHECK(hipGetLastError());
    ctx->last_w = width; ctx->last_h = height;
    return PT_OK;
}

int pt_trace_last_result(PtContext* ctx, float* result_host)
{
    PT_REQUIRE(ctx && result_host, "null argument");
    PT_REQUIRE(ctx->last_w && ctx->ws.frame_result, "no frame has been traced");
    int rc = pt_lanes_sync(ctx);
    if (rc) return rc;
    PT_HIP_CHECK(hipMemcpy(result_host, ctx->ws.frame_result, sizeof(float) * 3 * (size_t)ctx->last_w * ctx->last_h, hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_trace_stats(PtContext* ctx, PtTraceStats* out)
{
    PT_REQUIRE(ctx && out, "null argument");
    memset(out, 0, sizeof *out);
    if (!ctx->ws.stats) return PT_OK;
    int rc = pt_lanes_sync(ctx);
    if (rc) return rc;
    uint64_t h[PT_MAX_DEPTH + 1];
    PT_HIP_CHECK(hipMemcpy(h, ctx->ws.stats, sizeof h, hipMemcpyDeviceToHost));
    for (int d = 0; d < PT_MAX_DEPTH; ++d) { out->rays_per_depth[d] = h[d]; out->rays += h[d]; }
    out->paths = h[PT_MAX_DEPTH];
    return PT_OK;
}

int pt_ctx_set_share_primary(PtContext* ctx, int enabled)
{
    PT_REQUIRE(ctx, "null argument");
    ctx->share_primary = enabled != 0;     // read when a frame is enqueued: frames in flight keep what they were enqueued with
    return PT_OK;
}

int pt_ctx_set_profiling(PtContext* ctx, int enabled)
{
    PT_REQUIRE(ctx, "null argument");
    ctx->profiling = enabled != 0;
    return PT_OK;
}

int pt_trace_profile(PtContext* ctx, PtTraceProfile* out)
{
    PT_REQUIRE(ctx && out, "null argument");
    memset(out, 0, sizeof *out);
    int rc = pt_lanes_sync(ctx);
    if (rc) return rc;
    for (size_t k = 0; k + 1 < ctx->ev_used; k += 2) {
        float ms = 0.0f;
        PT_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev_pool[k], ctx->ev_pool[k + 1]));
        if (ctx->ev_primary[k / 2]) { out->primary_ms += ms; out->primary_launches++; }
        else { out->intersect_ms += ms; out->intersect_launches++; }
    }
    ctx->ev_used = 0;
    if (ctx->ws.stats) {
        uint64_t h[PT_MAX_DEPTH + 1];
        PT_HIP_CHECK(hipMemcpy(h, ctx->ws.stats, sizeof h, hipMemcpyDeviceToHost));
        for (int d = 0; d < PT_MAX_DEPTH; ++d) { (d == 0 ? out->primary_rays : out->intersect_rays) += h[d] - ctx->stats_mark[d]; ctx->stats_mark[d] = h[d]; }
    }
    return PT_OK;
}

int pt_trace_stats_reset(PtContext* ctx)
{
    PT_REQUIRE(ctx, "null argument");
    if (!ctx->ws.stats) return PT_OK;
    int rc = pt_lanes_sync(ctx);
    if (rc) return rc;
    PT_HIP_CHECK(hipMemsetAsync(ctx->ws.stats, 0, sizeof(uint64_t) * (PT_MAX_DEPTH + 1), ctx->stream));
    memset(ctx->stats_mark, 0, sizeof ctx->stats_mark);
    return PT_OK;
}

int pt_accum_resolve(PtContext* ctx, const float* accum_sum_dev, uint32_t n_pixels, uint8_t* rgba8_dev, int bgra)
{
    PT_REQUIRE(ctx && accum_sum_dev && rgba8_dev, "null argument");
    if (n_pixels == 0) return PT_OK;
    PT_HIP_CHECK(hipSetDevice(ctx->device));
    int rc = pt_lanes_join(ctx);   // frames in flight still add to the sum
    if (rc) return rc;
    k_resolve<<<(n_pixels + kBlock - 1) / kBlock, kBlock, 0, ctx->stream>>>(n_pixels, (const float4*)accum_sum_dev, (uchar4*)rgba8_dev, bgra);
    PT_HIP_CHECK(hipGetLastError());
    return PT_OK;
}

} // extern "C"

// ---- sum of the accumulation images of several contexts over peer access ------------------------
namespace {
constexpr uint32_t kMaxPeers = 16;
struct PtPeerImages { float4* img[kMaxPeers]; };

// out[i] = img[0][i] + img[1][i] + ... in this order, for i in [begin, end); out is img[self]
__global__ void k_sum_peers(PtPeerImages P, uint32_t n, uint32_t self, uint32_t begin, uint32_t end)
{
    const uint32_t i = begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= end) return;
    float4 a = P.img[0][i];
    for (uint32_t k = 1; k < n; ++k) { const float4 b = P.img[k][i]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    P.img[self][i] = a;
}
__global__ void k_copy_peer(const float4* __restrict__ src, float4* __restrict__ dst, uint32_t begin, uint32_t end)
{
    const uint32_t i = begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < end) dst[i] = src[i];
}
} // namespace

extern "C" int pt_accum_allreduce(PtContext* const* ctxs, void* const* accum, uint32_t n, uint32_t n_pixels)
{
    PT_REQUIRE(ctxs && accum && n >= 1 && n <= kMaxPeers, "1 to 16 contexts");
    for (uint32_t r = 0; r < n; ++r) PT_REQUIRE(ctxs[r] && accum[r], "null context or image");
    // images of contexts on one device must not overlap (device addresses of different GPUs are not comparable)
    for (uint32_t r = 0; r < n; ++r) for (uint32_t k = 0; k < r; ++k) {
        if (ctxs[r]->device != ctxs[k]->device) continue;
        const uintptr_t a = (uintptr_t)accum[r], b = (uintptr_t)accum[k], bytes = (uintptr_t)n_pixels * sizeof(float4);
        PT_REQUIRE(a != b && (a < b ? b - a : a - b) >= bytes, "two contexts' images overlap");
    }
    if (n == 1 || n_pixels == 0) return PT_OK;
    // peer access, both directions, once per pair of distinct devices
    for (uint32_t r = 0; r < n; ++r) {
        PT_HIP_CHECK(hipSetDevice(ctxs[r]->device));
        for (uint32_t k = 0; k < n; ++k) {
            if (ctxs[k]->device == ctxs[r]->device) continue;
            int can = 0;
            PT_HIP_CHECK(hipDeviceCanAccessPeer(&can, ctxs[r]->device, ctxs[k]->device));
            if (!can) { pt_set_error("pt_accum_allreduce: device %d cannot access device %d", ctxs[r]->device, ctxs[k]->device); return PT_ERROR_HIP; }
            const hipError_t e = hipDeviceEnablePeerAccess(ctxs[k]->device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { pt_set_error("hipDeviceEnablePeerAccess: %s", hipGetErrorString(e)); return PT_ERROR_HIP; }
            (void)hipGetLastError();
        }
    }
    PtPeerImages P;
    for (uint32_t k = 0; k < kMaxPeers; ++k) P.img[k] = k < n ? (float4*)accum[k] : nullptr;
    std::vector<hipEvent_t> ev(2 * n, nullptr);
    auto cleanup = [&]() { for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); };
    auto fail = [&](hipError_t e, const char* what) { pt_set_error("pt_accum_allreduce: %s: %s", what, hipGetErrorString(e)); cleanup(); return (int)PT_ERROR_HIP; };
    hipError_t e = hipSuccess;
    // every image is complete on its own stream ...
    for (uint32_t r = 0; r < n; ++r) {
        if ((e = hipSetDevice(ctxs[r]->device)) != hipSuccess) return fail(e, "hipSetDevice");
        int rc = pt_lanes_join(ctxs[r]);
        if (rc) { cleanup(); return rc; }
        // recorded with a release to system scope: what this GPU wrote (its image, then its slice of the sum) is visible
        // to the peers that pass the wait, whatever the platform's default event scope is
        if ((e = hipEventCreateWithFlags(&ev[r], hipEventDisableTiming | hipEventReleaseToSystem)) != hipSuccess) return fail(e, "hipEventCreate");
        if ((e = hipEventCreateWithFlags(&ev[n + r], hipEventDisableTiming | hipEventReleaseToSystem)) != hipSuccess) return fail(e, "hipEventCreate");
        if ((e = hipEventRecord(ev[r], ctxs[r]->stream)) != hipSuccess) return fail(e, "hipEventRecord");
    }
    // ... before anyone reads it: GPU r sums slice r
    const uint32_t per = (n_pixels + n - 1) / n;
    for (uint32_t r = 0; r < n; ++r) {
        if ((e = hipSetDevice(ctxs[r]->device)) != hipSuccess) return fail(e, "hipSetDevice");
        for (uint32_t k = 0; k < n; ++k) if (k != r && (e = hipStreamWaitEvent(ctxs[r]->stream, ev[k], 0)) != hipSuccess) return fail(e, "hipStreamWaitEvent");
        const uint32_t b = std::min(n_pixels, r * per), t = std::min(n_pixels, b + per);
        if (t > b) k_sum_peers<<<(t - b + kBlock - 1) / kBlock, kBlock, 0, ctxs[r]->stream>>>(P, n, r, b, t);
        if ((e = hipEventRecord(ev[n + r], ctxs[r]->stream)) != hipSuccess) return fail(e, "hipEventRecord");
    }
    // every GPU fetches the other slices from their owners
    for (uint32_t r = 0; r < n; ++r) {
        if ((e = hipSetDevice(ctxs[r]->device)) != hipSuccess) return fail(e, "hipSetDevice");
        for (uint32_t k = 0; k < n; ++k) if (k != r && (e = hipStreamWaitEvent(ctxs[r]->stream, ev[n + k], 0)) != hipSuccess) return fail(e, "hipStreamWaitEvent");
        for (uint32_t k = 0; k < n; ++k) {
            if (k == r) continue;
            const uint32_t b = std::min(n_pixels, k * per), t = std::min(n_pixels, b + per);
            if (t > b) k_copy_peer<<<(t - b + kBlock - 1) / kBlock, kBlock, 0, ctxs[r]->stream>>>(P.img[k], P.img[r], b, t);
        }
    }
    // a later call must not overwrite slice k of image k while GPU r still copies it: every stream waits for
    // every copy.  The events are recorded again for that and destroyed at once (the runtime keeps them alive
    // until they have completed).
    for (uint32_t r = 0; r < n; ++r) {
        if ((e = hipSetDevice(ctxs[r]->device)) != hipSuccess) return fail(e, "hipSetDevice");
        if ((e = hipEventRecord(ev[r], ctxs[r]->stream)) != hipSuccess) return fail(e, "hipEventRecord");
    }
    for (uint32_t r = 0; r < n; ++r) {
        if ((e = hipSetDevice(ctxs[r]->device)) != hipSuccess) return fail(e, "hipSetDevice");
        for (uint32_t k = 0; k < n; ++k) if (k != r && (e = hipStreamWaitEvent(ctxs[r]->stream, ev[k], 0)) != hipSuccess) return fail(e, "hipStreamWaitEvent");
    }
    e = hipGetLastError();
    if (e != hipSuccess) return fail(e, "kernel launch");
    cleanup();
    return PT_OK;
}

// ---- self-test of the short arithmetic forms ------------------------------------------------------
// pt_rcp, pt_sqrt, pt_div_pi and pt_unorm8 replace IEEE divisions and square roots on the strength of an exhaustive
// comparison; this entry point repeats that comparison on the GPU in use: every binary32 pattern through the public
// forms (guards included) against the compiler's IEEE expansions, every byte through pt_unorm8, and pt_div against the
// 32-bit division for a set of divisors and every third index below 2^31.
namespace {
__global__ void k_selftest_float(unsigned long long* __restrict__ bad)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long n_rcp = 0, n_sqrt = 0, n_pi = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const float x = __uint_as_float((uint32_t)i);
        const float a = pt_rcp(x), a0 = 1.0f / x;
        const float b = pt_sqrt(x), b0 = sqrtf(x);
        const float c = pt_div_pi(x), c0 = x / 3.14159265f;
        n_rcp += (__float_as_uint(a) != __float_as_uint(a0) && !(a != a && a0 != a0)) ? 1u : 0u;
        n_sqrt += (__float_as_uint(b) != __float_as_uint(b0) && !(b != b && b0 != b0)) ? 1u : 0u;
        n_pi += (__float_as_uint(c) != __float_as_uint(c0) && !(c != c && c0 != c0)) ? 1u : 0u;
    }
    if (n_rcp) atomicAdd(&bad[0], n_rcp);
    if (n_sqrt) atomicAdd(&bad[1], n_sqrt);
    if (n_pi) atomicAdd(&bad[2], n_pi);
    if (blockIdx.x == 0 && threadIdx.x < 256) {
        const float t = pt_unorm8(threadIdx.x), t0 = (float)threadIdx.x / 255.0f;
        if (__float_as_uint(t) != __float_as_uint(t0)) atomicAdd(&bad[3], 1ull);
    }
}
__global__ void k_selftest_div(PtDivisor D, unsigned long long* __restrict__ bad)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * 3u;
    unsigned long long n = 0;
    for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 3u; i < (1ull << 31); i += stride)
        n += pt_div((uint32_t)i, D) != (uint32_t)i / D.d ? 1u : 0u;
    // the ends of the range, whatever the stride
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        const uint32_t e = 0x7FFFFFFFu - threadIdx.x;
        n += pt_div(e, D) != e / D.d ? 1u : 0u;
        n += pt_div(threadIdx.x, D) != threadIdx.x / D.d ? 1u : 0u;
    }
    if (n) atomicAdd(&bad[4], n);
}
} // namespace

extern "C" int pt_selftest_arith(PtContext* ctx, uint64_t mismatches[5])
{
    PT_REQUIRE(ctx && mismatches, "null argument");
    PT_HIP_CHECK(hipSetDevice(ctx->device));
    unsigned long long* d = nullptr;
    PT_HIP_CHECK(hipMalloc((void**)&d, sizeof(unsigned long long) * 5));
    hipError_t e = hipMemsetAsync(d, 0, sizeof(unsigned long long) * 5, ctx->stream);
    if (e == hipSuccess) {
        k_selftest_float<<<ctx->num_cus * 8, kBlock, 0, ctx->stream>>>(d);
        const uint32_t divisors[] = { 1u, 2u, 3u, 5u, 7u, 16u, 25u, 64u, 255u, 256u, 641u, 1000u, 1080u, 1920u, 3840u, 65535u, 65536u, 0x7FFFFFFFu };
        for (uint32_t dv : divisors) k_selftest_div<<<ctx->num_cus * 8, kBlock, 0, ctx->stream>>>(pt_divisor(dv), d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(mismatches, d, sizeof(unsigned long long) * 5, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (e != hipSuccess) { pt_set_error("pt_selftest_arith: %s", hipGetErrorString(e)); return PT_ERROR_HIP; }
    return PT_OK;
}

#ifdef PT_SCHED_STATS
extern "C" int pt_debug_sched_stats(unsigned long long* out8, int reset)
{
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(pt_sched_stats), 64) != hipSuccess) return 1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(pt_sched_stats), z, 64) != hipSuccess) return 1; }
    return 0;
}
#endif
