// pt_build.hip -- BLAS / TLAS builders for gfx950.
//
// Replaces the driver BVH builds behind
//   source/bottom_level_acceleration_structure.cpp:74-77  (vkCmdBuildAccelerationStructuresKHR, BLAS)
//   source/top_level_acceleration_structure.cpp:109-112   (same, TLAS)
// with an LBVH: 63-bit Morton keys -> stable LSD radix sort -> Karras radix tree -> bottom-up
// boxes + area-weighted SAH cost with leaf collapse -> compact 64-byte nodes that carry both
// child boxes (one node fetch = four 16-byte loads per lane) and 48-byte triangle records in
// leaf order.  Every step is a pure function of its input, so the structure does not depend on
// scheduling; nothing here hands data between workgroups inside a launch except through
// commutative integer atomics read by a later kernel.
#include "pt_internal.h"
#include "pt_wide.h"
#include "pt_sah.h"
#include <math.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <limits>

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ uint32_t f2ord(float f)
{
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ __forceinline__ float ord2f(uint32_t u)
{
    uint32_t b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
#ifdef __HIP_DEVICE_COMPILE__
    return __uint_as_float(b);
#else
    float f; memcpy(&f, &b, 4); return f;
#endif
}

// ---------------------------------------------------------------- triangle boxes
__global__ void k_blas_maxabs(const uint8_t* __restrict__ vertices, uint32_t stride, const uint32_t* __restrict__ indices,
                              uint32_t n_prims, uint32_t* __restrict__ maxabs_bits)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    float m = 0.0f;
    if (i < n_prims) {
        for (int k = 0; k < 3; ++k) {
            const float* p = (const float*)(vertices + (size_t)indices[3 * i + k] * stride);
            m = fmaxf(m, fmaxf(fabsf(p[0]), fmaxf(fabsf(p[1]), fabsf(p[2]))));
        }
    }
    // non-negative floats order like their bit patterns
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) atomicMax(maxabs_bits, __float_as_uint(m));
}

__global__ void k_tri_boxes(const uint8_t* __restrict__ vertices, uint32_t stride, const uint32_t* __restrict__ indices,
                            uint32_t n_prims, const uint32_t* __restrict__ maxabs_bits, float* __restrict__ boxes,
                            uint32_t* __restrict__ root_ord /* 6: lo xyz (min), hi xyz (max) */)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n_prims;
    const uint32_t j = valid ? i : n_prims - 1;          // lanes past the end repeat the last primitive: min and max do not mind
    const float delta = __uint_as_float(*maxabs_bits) * 3.0517578125e-05f;
    const float* a = (const float*)(vertices + (size_t)indices[3 * j + 0] * stride);
    const float* b = (const float*)(vertices + (size_t)indices[3 * j + 1] * stride);
    const float* c = (const float*)(vertices + (size_t)indices[3 * j + 2] * stride);
    for (int k = 0; k < 3; ++k) {
        float lo = fminf(fminf(a[k], b[k]), c[k]) - delta;
        float hi = fmaxf(fmaxf(a[k], b[k]), c[k]) + delta;
        if (valid) { boxes[6 * (size_t)i + k] = lo; boxes[6 * (size_t)i + 3 + k] = hi; }
        // one pair of atomics per wave instead of per primitive
        uint32_t ul = f2ord(lo), uh = f2ord(hi);
        for (int off = 32; off > 0; off >>= 1) { ul = min(ul, (uint32_t)__shfl_xor(ul, off)); uh = max(uh, (uint32_t)__shfl_xor(uh, off)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&root_ord[k], ul); atomicMax(&root_ord[3 + k], uh); }
    }
}

// ---------------------------------------------------------------- Morton keys
__global__ void k_centroid_bounds(const float* __restrict__ boxes, uint32_t n, uint32_t* __restrict__ cb_ord)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int k = 0; k < 3; ++k) {
        float c = (boxes[6 * (size_t)i + k] + boxes[6 * (size_t)i + 3 + k]) * 0.5f;
        atomicMin(&cb_ord[k], f2ord(c));
        atomicMax(&cb_ord[3 + k], f2ord(c));
    }
}

__device__ __forceinline__ uint64_t expand21(uint32_t v)
{
    uint64_t x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
__device__ __forceinline__ uint32_t quantize21(float c, float cmin, float cmax)
{
    float ext = cmax - cmin;
    float scale = ext > 0.0f ? 2097152.0f / ext : 0.0f;
    float q = (c - cmin) * scale;
    q = fminf(fmaxf(q, 0.0f), 2097151.0f);
    return (uint32_t)q;
}

__global__ void k_morton(const float* __restrict__ boxes, uint32_t n, const uint32_t* __restrict__ cb_ord,
                         uint64_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t q[3];
    for (int k = 0; k < 3; ++k) {
        float c = (boxes[6 * (size_t)i + k] + boxes[6 * (size_t)i + 3 + k]) * 0.5f;
        q[k] = quantize21(c, ord2f(cb_ord[k]), ord2f(cb_ord[3 + k]));
    }
    keys[i] = (expand21(q[0]) << 2) | (expand21(q[1]) << 1) | expand21(q[2]);
    vals[i] = i;
}

// ---------------------------------------------------------------- stable LSD radix sort, 8-bit digits
constexpr int kSortItems = 16;                       // sub-tiles of 256 keys per block
constexpr int kSortTile = kBlock * kSortItems;

__global__ void k_sort_hist(const uint64_t* __restrict__ keys, uint32_t n, int shift, uint32_t n_blocks, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    size_t base = (size_t)blockIdx.x * kSortTile;
    for (int s = 0; s < kSortItems; ++s) {
        size_t idx = base + (size_t)s * kBlock + threadIdx.x;
        if (idx < n) atomicAdd(&h[(uint32_t)(keys[idx] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * n_blocks + blockIdx.x] = h[threadIdx.x];
}

__global__ void k_sort_scatter(const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t n, int shift,
                               uint32_t n_blocks, const uint32_t* __restrict__ hist_scanned,
                               uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out)
{
    __shared__ uint32_t base[256];
    __shared__ uint32_t wcount[4][256];
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    base[tid] = hist_scanned[(size_t)tid * n_blocks + blockIdx.x];
    size_t tile = (size_t)blockIdx.x * kSortTile;
    for (int s = 0; s < kSortItems; ++s) {
        size_t idx = tile + (size_t)s * kBlock + tid;
        bool valid = idx < n;
        uint64_t key = valid ? keys_in[idx] : 0;
        uint32_t val = valid ? vals_in[idx] : 0;
        uint32_t digit = (uint32_t)(key >> shift) & 255u;
        for (int w = 0; w < 4; ++w) wcount[w][tid] = 0;
        __syncthreads();
        // lanes of this wave holding the same digit
        unsigned long long mask = __ballot(valid);
        for (int b = 0; b < 8; ++b) {
            unsigned long long bal = __ballot((digit >> b) & 1u);
            mask &= ((digit >> b) & 1u) ? bal : ~bal;
        }
        uint32_t rank = __popcll(mask & ((1ull << lane) - 1ull));
        if (valid && rank == 0) wcount[wave][digit] = __popcll(mask);
        __syncthreads();
        if (valid) {
            uint32_t off = base[digit] + rank;
            for (uint32_t w = 0; w < wave; ++w) off += wcount[w][digit];
            keys_out[off] = key;
            vals_out[off] = val;
        }
        __syncthreads();
        base[tid] += wcount[0][tid] + wcount[1][tid] + wcount[2][tid] + wcount[3][tid];
        __syncthreads();
    }
}

// ---------------------------------------------------------------- exclusive scan (u32)
__global__ void k_scan_block(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t n, uint32_t* __restrict__ block_sums)
{
    __shared__ uint32_t wsum[4];
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    uint32_t v = i < n ? in[i] : 0;
    uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = v;
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t y = __shfl_up(x, off);
        if (lane >= (uint32_t)off) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    uint32_t add = 0;
    for (uint32_t w = 0; w < wave; ++w) add += wsum[w];
    if (i < n) out[i] = x - v + add;
    if (threadIdx.x == kBlock - 1) block_sums[blockIdx.x] = x + add;
}
// one block walks the block sums in order (n_blocks is small: n / 256)
__global__ void k_scan_sums(uint32_t* __restrict__ sums, uint32_t n_blocks, uint32_t* __restrict__ total)
{
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t start = 0; start < n_blocks; start += kBlock) {
        uint32_t i = start + threadIdx.x;
        uint32_t v = i < n_blocks ? sums[i] : 0;
        uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        uint32_t x = v;
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t y = __shfl_up(x, off);
            if (lane >= (uint32_t)off) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        uint32_t add = carry;
        for (uint32_t w = 0; w < wave; ++w) add += wsum[w];
        if (i < n_blocks) sums[i] = x - v + add;
        __syncthreads();
        if (threadIdx.x == kBlock - 1) carry = x + add;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total) __hip_atomic_store(total, carry, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_scan_add(uint32_t* __restrict__ out, uint32_t n, const uint32_t* __restrict__ block_sums)
{
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[i] += block_sums[blockIdx.x];
}
// The same with the block sums as k_scan_block left them: every block adds up the sums in front of it
// (a few hundred words for the arrays of a build), which saves the launch in between.
__global__ void k_scan_add_raw(uint32_t* __restrict__ out, uint32_t n, const uint32_t* __restrict__ block_sums, uint32_t n_blocks,
                               uint32_t* __restrict__ total)
{
    __shared__ uint32_t wsum[4];
    const bool last = blockIdx.x == n_blocks - 1;
    const uint32_t upto = (last && total) ? n_blocks : blockIdx.x;      // the last block also reports the grand total
    uint32_t mine = 0, own = 0;
    for (uint32_t k = threadIdx.x; k < upto; k += kBlock) { const uint32_t v = block_sums[k]; if (k < blockIdx.x) mine += v; else own += v; }
    for (int off = 32; off > 0; off >>= 1) { mine += __shfl_xor(mine, off); own += __shfl_xor(own, off); }
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = mine;
    __syncthreads();
    const uint32_t add = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[i] += add;
    if (last && total) {
        __syncthreads();
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = own;
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(total, add + wsum[0] + wsum[1] + wsum[2] + wsum[3], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---------------------------------------------------------------- Karras 2012 radix tree
__device__ __forceinline__ int lcp(const uint64_t* __restrict__ keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    uint64_t a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((uint32_t)i ^ (uint32_t)j);
    return __clzll((long long)(a ^ b));
}

// inner node i in [0, n-2]; child encoding: >= 0 inner index, < 0 : ~leaf position
__global__ void k_karras(const uint64_t* __restrict__ keys, int n, int* __restrict__ left, int* __restrict__ right,
                         uint32_t* __restrict__ rng_lo, int* __restrict__ parent)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    int d = (lcp(keys, n, i, i + 1) - lcp(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    int dmin = lcp(keys, n, i, i - d);
    int lmax = 2;
    while (lcp(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (lcp(keys, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = lcp(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (lcp(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    int gamma = i + s * d + min(d, 0);
    int lo = min(i, j), hi = max(i, j);
    rng_lo[i] = (uint32_t)lo;
    if (lo == gamma) left[i] = ~gamma; else { left[i] = gamma; parent[gamma] = i; }
    if (hi == gamma + 1) right[i] = ~(gamma + 1); else { right[i] = gamma + 1; parent[gamma + 1] = i; }
    if (i == 0) parent[0] = -1;
}

// ---------------------------------------------------------------- bottom-up passes
struct InnerData {       // per inner node
    float lo[3], hi[3];
    uint32_t count;
    float cost;
};

__device__ __forceinline__ float half_area(const float lo[3], const float hi[3])
{
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return __builtin_fmaf(dz, dx, __builtin_fmaf(dy, dz, dx * dy));
}

// One pass: every inner node whose children were finished in an earlier pass finishes now.
// done[i] = pass number (>0) in which node i was finished.  A child finished in THIS pass reads
// as unfinished whichever value the racing load returns, so the outcome is schedule independent.
__global__ void k_refit_pass(int n_inner, uint32_t pass, const int* __restrict__ left, const int* __restrict__ right,
                             const float* __restrict__ boxes, const uint32_t* __restrict__ order,
                             InnerData* __restrict__ data, uint8_t* __restrict__ collapsed, uint32_t* __restrict__ done,
                             uint32_t max_leaf, uint32_t* __restrict__ n_done)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_inner) return;
    if (done[i]) return;
    int l = left[i], r = right[i];
    if (l >= 0) { uint32_t dl = done[l]; if (dl == 0 || dl >= pass) return; }
    if (r >= 0) { uint32_t dr = done[r]; if (dr == 0 || dr >= pass) return; }
    float llo[3], lhi[3], rlo[3], rhi[3], lcost, rcost; uint32_t lc, rc;
    if (l >= 0) { for (int k = 0; k < 3; ++k) { llo[k] = data[l].lo[k]; lhi[k] = data[l].hi[k]; } lc = data[l].count; lcost = data[l].cost; }
    else { const float* b = boxes + 6 * (size_t)order[~l]; for (int k = 0; k < 3; ++k) { llo[k] = b[k]; lhi[k] = b[3 + k]; } lc = 1; lcost = half_area(llo, lhi); }
    if (r >= 0) { for (int k = 0; k < 3; ++k) { rlo[k] = data[r].lo[k]; rhi[k] = data[r].hi[k]; } rc = data[r].count; rcost = data[r].cost; }
    else { const float* b = boxes + 6 * (size_t)order[~r]; for (int k = 0; k < 3; ++k) { rlo[k] = b[k]; rhi[k] = b[3 + k]; } rc = 1; rcost = half_area(rlo, rhi); }
    InnerData d;
    for (int k = 0; k < 3; ++k) { d.lo[k] = fminf(llo[k], rlo[k]); d.hi[k] = fmaxf(lhi[k], rhi[k]); }
    d.count = lc + rc;
    float A = half_area(d.lo, d.hi);
    float cint = A + (lcost + rcost);
    float cleaf = (float)d.count * A;
    bool col = (i != 0) && d.count <= max_leaf && cleaf <= cint;
    d.cost = col ? cleaf : cint;
    data[i] = d;
    collapsed[i] = col ? 1 : 0;
    done[i] = pass;
    atomicAdd(n_done, 1u);
}

__global__ void k_alive(int n_inner, const int* __restrict__ parent, const InnerData* __restrict__ data,
                        const uint8_t* __restrict__ collapsed, uint32_t max_leaf, uint32_t* __restrict__ alive)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_inner) return;
    bool dead = collapsed[i] != 0;
    for (int p = parent[i]; !dead && p >= 0 && data[p].count <= max_leaf; p = parent[p])
        if (collapsed[p]) dead = true;
    alive[i] = dead ? 0u : 1u;
}

__global__ void k_emit(int n_inner, const int* __restrict__ left, const int* __restrict__ right, const uint32_t* __restrict__ rng_lo,
                       const float* __restrict__ boxes, const uint32_t* __restrict__ order, const InnerData* __restrict__ data,
                       const uint8_t* __restrict__ collapsed, const uint32_t* __restrict__ alive, const uint32_t* __restrict__ new_index,
                       PtBvhNode* __restrict__ nodes)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_inner || !alive[i]) return;
    PtBvhNode nd;
    nd.aux0 = nd.aux1 = 0;
    int ch[2] = { left[i], right[i] };
    for (int s = 0; s < 2; ++s) {
        int c = ch[s];
        float* lo = s == 0 ? nd.lo0 : nd.lo1;
        float* hi = s == 0 ? nd.hi0 : nd.hi1;
        int ref;
        if (c < 0) {
            const float* b = boxes + 6 * (size_t)order[~c];
            for (int k = 0; k < 3; ++k) { lo[k] = b[k]; hi[k] = b[3 + k]; }
            ref = ~(int)(((uint32_t)(~c) << 3) | 0u);
        } else {
            for (int k = 0; k < 3; ++k) { lo[k] = data[c].lo[k]; hi[k] = data[c].hi[k]; }
            ref = collapsed[c] ? ~(int)((rng_lo[c] << 3) | (data[c].count - 1u)) : (int)new_index[c];
        }
        if (s == 0) nd.c0 = ref; else nd.c1 = ref;
    }
    nodes[new_index[i]] = nd;
}

// 48-byte triangle record per sorted slot: (v0.xyz, v1.x) (v1.yz, v2.xy) (v2.z, bits(prim), 0, 0)
__global__ void k_pack_tris(const uint8_t* __restrict__ vertices, uint32_t stride, const uint32_t* __restrict__ indices,
                            const uint32_t* __restrict__ order, uint32_t n, float4* __restrict__ tris)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t p = order[i];
    const float* a = (const float*)(vertices + (size_t)indices[3 * p + 0] * stride);
    const float* b = (const float*)(vertices + (size_t)indices[3 * p + 1] * stride);
    const float* c = (const float*)(vertices + (size_t)indices[3 * p + 2] * stride);
    tris[3 * (size_t)i + 0] = make_float4(a[0], a[1], a[2], b[0]);
    tris[3 * (size_t)i + 1] = make_float4(b[1], b[2], c[0], c[1]);
    tris[3 * (size_t)i + 2] = make_float4(c[2], __uint_as_float(p), 0.0f, 0.0f);
}

// ---- traversal layout on the device: one breadth-first level per pass (pt_wide.h has the per-node routines)
__global__ void k_wide_count(const PtBvhNode* __restrict__ n2, const uint32_t* __restrict__ frontier, uint32_t n_front,
                             uint32_t* __restrict__ counts)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_front) return;
    PtWideEntry ent[4];
    const int ne = pt_wide_open(n2, frontier[i], ent);
    uint32_t c = 0;
    for (int e = 0; e < ne; ++e) c += ent[e].ref >= 0 ? 1u : 0u;
    counts[i] = c;
}

// Wide node level_base + i stands for frontier[i]; its inner entries become the next level in
// frontier order (offsets = exclusive scan of the counts), which is the numbering of a host
// breadth-first walk.
__global__ void k_wide_emit(const PtBvhNode* __restrict__ n2, const uint32_t* __restrict__ frontier, uint32_t n_front,
                            const uint32_t* __restrict__ offsets, uint32_t level_base, PtWide4* __restrict__ out,
                            uint32_t* __restrict__ next_frontier)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_front) return;
    PtWideEntry ent[4];
    const int ne = pt_wide_open(n2, frontier[i], ent);
    PtWide4 w;
    w.origin[0] = w.origin[1] = w.origin[2] = 0.0f; w.scale_x = w.scale_y = w.scale_z = 0.0f;
    for (int k = 0; k < 6; ++k) w.q[k] = 0u;
    pt_wide_quantise(ent, ne, w);
    uint32_t k = offsets[i];
    for (int e = 0; e < 4; ++e) {
        if (e >= ne) { w.child[e] = PT_WIDE_EMPTY; continue; }
        if (ent[e].ref >= 0) {
            w.child[e] = (int32_t)(level_base + n_front + k);
            next_frontier[k] = (uint32_t)ent[e].ref;
            ++k;
        } else w.child[e] = ent[e].ref;
    }
    out[level_base + i] = w;
}

// The pool's nodes with their planes expanded to floats by the arithmetic the quantised traversal step uses.
__global__ void k_wide_expand(const PtWide4* __restrict__ in, uint32_t n, PtWideF* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const PtWide4 w = in[i];
    PtWideF f;
    const float inf = __builtin_inff();
    for (int e = 0; e < 4; ++e) {
        const bool empty = pt_wide_slot_unused(w, e);
        f.lo_x[e] = empty ? inf : pt_grid_plane((w.q[0] >> (8 * e)) & 255u, w.scale_x, w.origin[0]);
        f.lo_y[e] = empty ? inf : pt_grid_plane((w.q[1] >> (8 * e)) & 255u, w.scale_y, w.origin[1]);
        f.lo_z[e] = empty ? inf : pt_grid_plane((w.q[2] >> (8 * e)) & 255u, w.scale_z, w.origin[2]);
        f.hi_x[e] = empty ? -inf : pt_grid_plane((w.q[3] >> (8 * e)) & 255u, w.scale_x, w.origin[0]);
        f.hi_y[e] = empty ? -inf : pt_grid_plane((w.q[4] >> (8 * e)) & 255u, w.scale_y, w.origin[1]);
        f.hi_z[e] = empty ? -inf : pt_grid_plane((w.q[5] >> (8 * e)) & 255u, w.scale_z, w.origin[2]);
        f.child[e] = w.child[e];
        f.pad[e] = 0u;
    }
    out[i] = f;
}

// Copies wide nodes into a pool: inner references move by node_base, triangle positions by tri_base.
// In the pool an unused slot is an inverted box (low bytes 255, high bytes 0: the far plane is never behind the
// near one, so no ray interval overlaps it) whose reference repeats slot 0's.  The traversal step then needs no test
// of the reference: should rounding ever collapse the six distances of such a slot into one value and let it
// pass, the ray visits slot 0's child a second time, which changes no closest hit.
__global__ void k_wide_rebase(const PtWide4* __restrict__ in, uint32_t n, uint32_t node_base, uint32_t tri_base, PtWide4* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    PtWide4 w = in[i];
    for (int e = 0; e < 4; ++e) {
        const int32_t c = w.child[e];
        if (c == PT_WIDE_EMPTY) {
            w.child[e] = w.child[0];       // (already rebased; slot 0 is always in use)
            for (int k = 0; k < 3; ++k) { w.q[k] |= 255u << (8 * e); w.q[3 + k] &= ~(255u << (8 * e)); }
            continue;
        }
        w.child[e] = c >= 0 ? (int32_t)((uint32_t)c + node_base) : pt_wide_leaf_rebased(c, tri_base);
    }
    out[node_base + i] = w;
}

// ---- binned-SAH tree for the traversal layout, one level per pass (pt_sah.h has the decisions)
#define PT_SAH_NONE 0xFFFFFFFFu
struct SahBinsOrd {                 // bins of one range as ordered uints, so that min/max are integer atomics
    uint32_t lo[3][PT_SAH_BINS][3], hi[3][PT_SAH_BINS][3], cnt[3][PT_SAH_BINS];
};

__global__ void k_iota(uint32_t* __restrict__ out, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = i;
}

__global__ void k_sah_root(const uint32_t* __restrict__ root_ord, uint32_t n, PtSahTask* __restrict__ tasks)
{
    PtSahTask t;
    memset(&t, 0, sizeof t);
    t.a = 0; t.b = n; t.parent = -1;
    for (int k = 0; k < 3; ++k) { t.lo[k] = ord2f(root_ord[k]); t.hi[k] = ord2f(root_ord[3 + k]); }
    tasks[0] = t;
}

__global__ void k_sah_reset(uint32_t m, SahBinsOrd* __restrict__ bins)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t per = sizeof(SahBinsOrd) / 4;
    if (i < m * per) {
        const uint32_t w = i % per;
        uint32_t* p = (uint32_t*)bins + i;
        const uint32_t n_box = 3 * PT_SAH_BINS * 3;
        *p = w < n_box ? 0xFFFFFFFFu : 0u;        // lo: identity of min; hi and counts: identity of max / sum
    }
}

// Positions are contiguous per range, so most of a block's primitives belong to the range of its first
// one: those are binned in LDS and flushed once (a few hundred global atomics per block instead of 21
// per primitive); the others go to global memory directly.
__global__ void __launch_bounds__(kBlock)
k_sah_bins(uint32_t n, const uint32_t* __restrict__ ids, const float* __restrict__ boxes,
           const uint32_t* __restrict__ task_of, uint32_t t_lo, uint32_t t_hi, const PtSahTask* __restrict__ tasks,
           SahBinsOrd* __restrict__ bins)
{
    constexpr uint32_t kWords = sizeof(SahBinsOrd) / 4, kBoxWords = 3 * PT_SAH_BINS * 3;
    __shared__ SahBinsOrd s_bins;
    __shared__ uint32_t s_task;
    uint32_t* sw = (uint32_t*)&s_bins;
    for (uint32_t w = threadIdx.x; w < kWords; w += kBlock) sw[w] = w < kBoxWords ? 0xFFFFFFFFu : 0u;
    const uint32_t first = blockIdx.x * blockDim.x;
    if (threadIdx.x == 0) s_task = first < n ? task_of[first] : PT_SAH_NONE;
    __syncthreads();
    uint32_t t0 = s_task;
    if (t0 != PT_SAH_NONE && (t0 < t_lo || t0 >= t_hi)) t0 = PT_SAH_NONE;
    const uint32_t i = first + threadIdx.x;
    if (i < n) {
        const uint32_t t = task_of[i];
        if (t != PT_SAH_NONE && t >= t_lo && t < t_hi) {
            const float* x = boxes + 6 * (size_t)ids[i];
            SahBinsOrd& B = t == t0 ? s_bins : bins[t - t_lo];
            for (int ax = 0; ax < 3; ++ax) {
                const float clo = tasks[t].lo[ax], chi = tasks[t].hi[ax];     // bins span the range's box
                if (!(chi > clo)) continue;
                const int q = pt_sah_bin_of(x, ax, clo, pt_sah_scale(clo, chi));
                atomicAdd(&B.cnt[ax][q], 1u);
                for (int k = 0; k < 3; ++k) { atomicMin(&B.lo[ax][q][k], f2ord(x[k])); atomicMax(&B.hi[ax][q][k], f2ord(x[3 + k])); }
            }
        }
    }
    __syncthreads();
    if (t0 == PT_SAH_NONE) return;
    uint32_t* gw = (uint32_t*)&bins[t0 - t_lo];
    for (uint32_t w = threadIdx.x; w < kWords; w += kBlock) {
        const uint32_t v = sw[w];
        if (w < kBoxWords) { if (v != 0xFFFFFFFFu) atomicMin(&gw[w], v); }
        else if (w < 2 * kBoxWords) { if (v != 0u) atomicMax(&gw[w], v); }
        else if (v != 0u) atomicAdd(&gw[w], v);
    }
}

// One wave per range.  Lane 16 * axis + q holds bin q of that axis; prefix and suffix unions run along
// the 16 lanes of an axis, every lane prices the split after its bin, and the cheapest split with the
// lowest (axis, bin) wins -- the first minimum of the serial search in pt_sah.h, whose second half
// (pt_sah_finish) lane 0 then runs on the winner.
__global__ void __launch_bounds__(64)
k_sah_decide(uint32_t t_lo, uint32_t t_hi, PtSahTask* __restrict__ tasks,
             SahBinsOrd* __restrict__ bins, uint32_t max_leaf, const float* __restrict__ boxes,
             const uint32_t* __restrict__ ids, uint32_t* __restrict__ inner)
{
    const uint32_t t = t_lo + blockIdx.x;
    if (t >= t_hi) return;
    const uint32_t lane = threadIdx.x;
    const int ax = (int)(lane >> 4), q = (int)(lane & 15u);
    const bool bin_lane = lane < 48u;
    SahBinsOrd& O = bins[t - t_lo];
    const float inf = __builtin_inff();
    const PtSahTask* T = &tasks[t];
    float c6[6];
    for (int k = 0; k < 3; ++k) { c6[k] = T->lo[k]; c6[3 + k] = T->hi[k]; }
    const uint32_t n = T->b - T->a;
    // this lane's bin; an empty bin still holds the identities, which do not decode to +-inf
    uint32_t cnt = 0;
    float lo[3] = { inf, inf, inf }, hi[3] = { -inf, -inf, -inf };
    if (bin_lane) {
        cnt = O.cnt[ax][q];
        if (cnt) for (int k = 0; k < 3; ++k) { lo[k] = ord2f(O.lo[ax][q][k]); hi[k] = ord2f(O.hi[ax][q][k]); }
        // the bins go back to the identities for the range that uses them next (k_sah_reset runs once per build)
        O.cnt[ax][q] = 0u;
        for (int k = 0; k < 3; ++k) { O.lo[ax][q][k] = 0xFFFFFFFFu; O.hi[ax][q][k] = 0u; }
    }
    // inclusive prefix (bins 0..q) and suffix (bins q..15) along the axis
    uint32_t pn = cnt, sn = cnt;
    float pl[3], ph[3], sl[3], sh[3];
    for (int k = 0; k < 3; ++k) { pl[k] = lo[k]; ph[k] = hi[k]; sl[k] = lo[k]; sh[k] = hi[k]; }
    for (int off = 1; off < 16; off <<= 1) {
        const uint32_t un = __shfl_up(pn, off, 16), dn = __shfl_down(sn, off, 16);
        float ul[3], uh[3], dl[3], dh[3];
        for (int k = 0; k < 3; ++k) {
            ul[k] = __shfl_up(pl[k], off, 16); uh[k] = __shfl_up(ph[k], off, 16);
            dl[k] = __shfl_down(sl[k], off, 16); dh[k] = __shfl_down(sh[k], off, 16);
        }
        if (q >= off) { pn += un; for (int k = 0; k < 3; ++k) { pl[k] = fminf(pl[k], ul[k]); ph[k] = fmaxf(ph[k], uh[k]); } }
        if (q + off < 16) { sn += dn; for (int k = 0; k < 3; ++k) { sl[k] = fminf(sl[k], dl[k]); sh[k] = fmaxf(sh[k], dh[k]); } }
    }
    // the split after bin q: left = bins 0..q, right = bins q+1..15 (the next lane's suffix)
    const uint32_t rn = __shfl_down(sn, 1, 16);
    float rl[3], rh[3];
    for (int k = 0; k < 3; ++k) { rl[k] = __shfl_down(sl[k], 1, 16); rh[k] = __shfl_down(sh[k], 1, 16); }
    float cost = inf;
    if (bin_lane && n > 1 && q < PT_SAH_BINS - 1 && c6[3 + ax] > c6[ax] && pn != 0 && rn != 0) {
        const float c = pt_sah_area(pl, ph) * (float)pn + pt_sah_area(rl, rh) * (float)rn;
        if (c < inf) cost = c;
    }
    // cheapest split, lowest lane on ties
    float bc = cost; uint32_t bl = lane;
    for (int off = 32; off > 0; off >>= 1) {
        const float oc = __shfl_xor(bc, off); const uint32_t ol = __shfl_xor(bl, off);
        if (oc < bc || (oc == bc && ol < bl)) { bc = oc; bl = ol; }
    }
    float w[12];
    for (int k = 0; k < 3; ++k) { w[k] = __shfl(pl[k], bl); w[3 + k] = __shfl(ph[k], bl); w[6 + k] = __shfl(rl[k], bl); w[9 + k] = __shfl(rh[k], bl); }
    if (lane != 0) return;
    PtSahTask task = *T;
    const bool have = bc < inf;
    pt_sah_finish(task, have ? (int)(bl >> 4) : -1, (int)(bl & 15u), bc, w, w + 3, w + 6, w + 9, c6, max_leaf);
    if (task.kind == 2) pt_sah_positional_boxes(task, boxes, ids);
    tasks[t] = task;
    inner[t] = task.kind != 0 ? 1u : 0u;
}

// The side of its range's split every primitive goes to.  Thread t < m also numbers range t if it is an inner one
// (offsets = exclusive scan of `inner`), writes its node and the reference its parent holds; that touches the
// range's `node` field only, which nothing else here reads.
__global__ void k_sah_emit_flags(uint32_t n, const uint32_t* __restrict__ ids, const float* __restrict__ boxes,
                                 const uint32_t* __restrict__ task_of, PtSahTask* __restrict__ tasks, uint32_t* __restrict__ flags,
                                 uint32_t m, const uint32_t* __restrict__ offsets, uint32_t node_base, PtBvhNode* __restrict__ nodes)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const uint32_t t = task_of[i];
        uint32_t f = 0;
        if (t != PT_SAH_NONE && tasks[t].kind != 0) f = pt_sah_goes_left(tasks[t], boxes + 6 * (size_t)ids[i], i) ? 1u : 0u;
        flags[i] = f;
    }
    if (i < m) {
        PtSahTask& task = tasks[i];
        int32_t ref;
        if (task.kind == 0) ref = pt_sah_leaf_ref(task.a, task.b - task.a);
        else {
            const uint32_t idx = node_base + offsets[i];
            task.node = idx;
            PtBvhNode& nd = nodes[idx];
            for (int k = 0; k < 3; ++k) { nd.lo0[k] = task.lo0[k]; nd.hi0[k] = task.hi0[k]; nd.lo1[k] = task.lo1[k]; nd.hi1[k] = task.hi1[k]; }
            nd.aux0 = nd.aux1 = 0;
            ref = (int32_t)idx;
        }
        if (task.parent >= 0) { if (task.side) nodes[task.parent].c1 = ref; else nodes[task.parent].c0 = ref; }
    }
}

// Stable partition of every split range (left[] = exclusive scan of flags over all positions) and the
// ranges of the next level: inner range number o of this level becomes ranges 2o and 2o + 1.
__global__ void k_sah_scatter(uint32_t n, const uint32_t* __restrict__ ids, const uint32_t* __restrict__ task_of,
                              const PtSahTask* __restrict__ tasks, const uint32_t* __restrict__ offsets,
                              const uint32_t* __restrict__ flags, const uint32_t* __restrict__ left,
                              uint32_t* __restrict__ ids_out, uint32_t* __restrict__ task_of_out, PtSahTask* __restrict__ next)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t t = task_of[i];
    if (t == PT_SAH_NONE || tasks[t].kind == 0) { ids_out[i] = ids[i]; task_of_out[i] = PT_SAH_NONE; return; }
    const PtSahTask& task = tasks[t];
    const uint32_t a = task.a, b = task.b;
    const uint32_t nl = left[b - 1] + flags[b - 1] - left[a];
    const uint32_t before = left[i] - left[a];
    const bool f = flags[i] != 0;
    const uint32_t pos = f ? a + before : a + nl + (i - a - before);
    ids_out[pos] = ids[i];
    task_of_out[pos] = 2u * offsets[t] + (f ? 0u : 1u);
    if (i == a) {
        PtSahTask c0, c1;
        memset(&c0, 0, sizeof c0); memset(&c1, 0, sizeof c1);
        c0.a = a; c0.b = a + nl; c0.parent = (int32_t)task.node; c0.side = 0;
        c1.a = a + nl; c1.b = b; c1.parent = (int32_t)task.node; c1.side = 1;
        for (int k = 0; k < 3; ++k) { c0.lo[k] = task.lo0[k]; c0.hi[k] = task.hi0[k]; c1.lo[k] = task.lo1[k]; c1.hi[k] = task.hi1[k]; }
        next[2u * offsets[t]] = c0; next[2u * offsets[t] + 1u] = c1;
    }
}

inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// Scratch memory of the builders.  A build makes some fifty short-lived device allocations, and a
// hipMalloc / hipFree pair costs more than most of the kernels in between, so they are carved from
// an arena the context keeps between builds (builds are synchronous on the context's stream, one
// at a time).  Nested builders release in stack order: a DevMem returns the arena to where it
// found it.  alloc_own is a plain hipMalloc for buffers that outlive the build.
struct DevMem {
    PtArena& arena;
    size_t mark_chunk, mark_used;
    std::vector<void*> own;
    explicit DevMem(PtContext* ctx) : arena(ctx->build_arena), mark_chunk(ctx->build_arena.cur), mark_used(ctx->build_arena.used) {}
    template <typename T> hipError_t alloc(T** p, size_t count)
    {
        const size_t bytes = ((count * sizeof(T) > 0 ? count * sizeof(T) : 16) + 255) & ~(size_t)255;
        while (arena.cur < arena.chunks.size() && arena.used + bytes > arena.chunks[arena.cur].second) { ++arena.cur; arena.used = 0; }
        if (arena.cur == arena.chunks.size()) {
            const size_t cap = bytes > ((size_t)32 << 20) ? bytes : ((size_t)32 << 20);
            void* q = nullptr;
            hipError_t e = hipMalloc(&q, cap);
            if (e != hipSuccess) return e;
            arena.chunks.push_back({ (char*)q, cap });
            arena.used = 0;
        }
        *p = (T*)(arena.chunks[arena.cur].first + arena.used);
        arena.used += bytes;
        return hipSuccess;
    }
    template <typename T> hipError_t alloc_own(T** p, size_t count)
    {
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, count * sizeof(T) > 0 ? count * sizeof(T) : 16);
        if (e == hipSuccess) { own.push_back(q); *p = (T*)q; }
        return e;
    }
    void release(void* p) { for (auto& q : own) if (q == p) q = nullptr; }
    ~DevMem()
    {
        for (void* q : own) if (q) (void)hipFree(q);
        arena.cur = mark_chunk; arena.used = mark_used;
        // back at the start with several chunks: one chunk of their total size serves the next build whole
        if (mark_chunk == 0 && mark_used == 0 && arena.chunks.size() > 1) {
            size_t total = 0;
            for (auto& c : arena.chunks) { total += c.second; (void)hipFree(c.first); }
            arena.chunks.clear();
            void* q = nullptr;
            if (hipMalloc(&q, total) == hipSuccess) arena.chunks.push_back({ (char*)q, total });
        }
    }
};

// One word back from the device: through the context's pinned word, so the copy is a DMA and not a
// staged pageable copy (the builders do this once per tree level).
hipError_t read_word(PtContext* ctx, const uint32_t* dev, uint32_t* out)
{
    if (!ctx->pinned_word) {
        hipError_t e = hipHostMalloc((void**)&ctx->pinned_word, 64, hipHostMallocDefault);
        if (e != hipSuccess) return e;
    }
    hipError_t e = hipMemcpyAsync(ctx->pinned_word, dev, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) *out = *ctx->pinned_word;
    return e;
}

// A per-level total without a stream synchronisation: the scan's last kernel stores it into a mapped host word
// that the host has set to an impossible value beforehand and then polls (a few microseconds after the store
// instead of a copy and a wait for the stream; the builders do this once per tree level).  Totals are counts of
// nodes or primitives, below 2^31.
constexpr uint32_t kMailboxEmpty = 0xFFFFFFFFu;
hipError_t mailbox_arm(PtContext* ctx, int slot, uint32_t** dev_word)
{
    if (getenv("PT_HIP_NO_MAILBOX")) return hipErrorNotSupported;      // for testing the other path
    if (!ctx->mailbox) {
        hipError_t e = hipHostMalloc((void**)&ctx->mailbox, 256, hipHostMallocMapped | hipHostMallocCoherent);
        if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&ctx->mailbox_dev, ctx->mailbox, 0);
        if (e != hipSuccess) { if (ctx->mailbox) (void)hipHostFree(ctx->mailbox); ctx->mailbox = nullptr; return e; }
    }
    __atomic_store_n(&ctx->mailbox[slot], kMailboxEmpty, __ATOMIC_RELEASE);
    *dev_word = ctx->mailbox_dev + slot;
    return hipSuccess;
}
hipError_t mailbox_wait(PtContext* ctx, int slot, uint32_t* out)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spin = 0;; ++spin) {
        const uint32_t v = __atomic_load_n(&ctx->mailbox[slot], __ATOMIC_ACQUIRE);
        if (v != kMailboxEmpty) { *out = v; return hipSuccess; }
        if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;
    }
    // not seen in time (a stalled queue, or a failed launch): wait for the stream and look once more
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return e;
    const uint32_t v = __atomic_load_n(&ctx->mailbox[slot], __ATOMIC_ACQUIRE);
    if (v == kMailboxEmpty) return hipErrorUnknown;
    *out = v;
    return hipSuccess;
}

// One per-level count: through the mailbox where mapped host memory is to be had, else through a device word,
// a copy and a wait for the stream.
struct LevelCount {
    PtContext* ctx; int slot; uint32_t* dev_fallback; bool polled = false;
    LevelCount(PtContext* c, int s, uint32_t* fallback) : ctx(c), slot(s), dev_fallback(fallback) {}
    uint32_t* arm()
    {
        uint32_t* w = nullptr;
        polled = mailbox_arm(ctx, slot, &w) == hipSuccess;
        if (!polled) (void)hipGetLastError();
        return polled ? w : dev_fallback;
    }
    hipError_t wait(uint32_t* out) { return polled ? mailbox_wait(ctx, slot, out) : read_word(ctx, dev_fallback, out); }
};

int scan_exclusive(hipStream_t st, DevMem& mem, const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* total_dev, uint32_t* sums = nullptr)
{
    uint32_t nb = div_up(n, kBlock);
    if (!sums) PT_HIP_CHECK(mem.alloc(&sums, nb));   // callers in a loop pass their own div_up(n, 256) words
    k_scan_block<<<nb, kBlock, 0, st>>>(in, out, n, sums);
    if (nb <= 4096) k_scan_add_raw<<<nb, kBlock, 0, st>>>(out, n, sums, nb, total_dev);
    else {
        k_scan_sums<<<1, kBlock, 0, st>>>(sums, nb, total_dev);
        k_scan_add<<<nb, kBlock, 0, st>>>(out, n, sums);
    }
    PT_HIP_CHECK(hipGetLastError());
    return PT_OK;
}

int radix_sort_pairs(hipStream_t st, DevMem& mem, uint64_t** keys, uint32_t** vals, uint32_t n, int key_bits)
{
    uint64_t* keys_alt = nullptr; uint32_t* vals_alt = nullptr; uint32_t* hist = nullptr; uint32_t* hist_scanned = nullptr;
    uint32_t nb = div_up(n, kSortTile);
    PT_HIP_CHECK(mem.alloc(&keys_alt, n));
    PT_HIP_CHECK(mem.alloc(&vals_alt, n));
    PT_HIP_CHECK(mem.alloc(&hist, (size_t)256 * nb));
    PT_HIP_CHECK(mem.alloc(&hist_scanned, (size_t)256 * nb));
    uint64_t* kin = *keys; uint32_t* vin = *vals; uint64_t* kout = keys_alt; uint32_t* vout = vals_alt;
    for (int shift = 0; shift < key_bits; shift += 8) {
        k_sort_hist<<<nb, kBlock, 0, st>>>(kin, n, shift, nb, hist);
        int rc = scan_exclusive(st, mem, hist, hist_scanned, 256 * nb, nullptr);
        if (rc) return rc;
        k_sort_scatter<<<nb, kBlock, 0, st>>>(kin, vin, n, shift, nb, hist_scanned, kout, vout);
        std::swap(kin, kout); std::swap(vin, vout);
    }
    PT_HIP_CHECK(hipGetLastError());
    *keys = kin; *vals = vin;
    return PT_OK;
}

} // namespace

// Binned-SAH binary tree over the padded primitive boxes (n > 1).  nodes_out[0] is the root;
// order_out[p] = primitive at leaf position p.  Both arrive pointing at the caller's buffers (n nodes, n words).
int pt_build_sah(PtContext* ctx, const float* boxes, uint32_t n, uint32_t max_leaf, const uint32_t* root_ord,
                 PtBvhNode** nodes_out, uint32_t* node_count_out, uint32_t** order_out)
{
    hipStream_t st = ctx->stream;
    DevMem mem(ctx);
    const uint32_t kBatch = 32768;                      // ranges whose bins are resident at once (44 MB)
    uint32_t *ids_a = nullptr, *ids_b = nullptr, *tof_a = nullptr, *tof_b = nullptr, *flags = nullptr, *left = nullptr;
    uint32_t *inner = nullptr, *offsets = nullptr, *total_dev = nullptr;
    PtSahTask *tasks_a = nullptr, *tasks_b = nullptr; SahBinsOrd* bins = nullptr; PtBvhNode* nodes = *nodes_out; uint32_t* sums = nullptr;
    PT_REQUIRE(nodes && *order_out, "pt_build_sah: output buffers missing");
    hipError_t e = hipSuccess;
    if ((e = mem.alloc(&ids_a, n)) || (e = mem.alloc(&ids_b, n)) || (e = mem.alloc(&tof_a, n)) || (e = mem.alloc(&tof_b, n)) ||
        (e = mem.alloc(&flags, n)) || (e = mem.alloc(&left, n)) || (e = mem.alloc(&inner, n)) || (e = mem.alloc(&offsets, n)) ||
        (e = mem.alloc(&tasks_a, n)) || (e = mem.alloc(&tasks_b, n)) ||
        (e = mem.alloc(&bins, kBatch)) || (e = mem.alloc(&sums, div_up(n, kBlock) + 1)) || (e = mem.alloc(&total_dev, 4))) {
        pt_set_error("pt_build_sah: hipMalloc failed: %s", hipGetErrorString(e)); return PT_ERROR_OUT_OF_MEMORY;
    }
    LevelCount count(ctx, 0, total_dev);
    const uint32_t nb = div_up(n, kBlock);
    k_iota<<<nb, kBlock, 0, st>>>(ids_a, n);
    PT_HIP_CHECK(hipMemsetAsync(tof_a, 0, sizeof(uint32_t) * n, st));          // every primitive in range 0
    k_sah_root<<<1, 1, 0, st>>>(root_ord, n, tasks_a);
    {   // identities in every bin once; k_sah_decide restores them after reading
        const uint32_t mb = n < kBatch ? n : kBatch;
        k_sah_reset<<<div_up(mb * (uint32_t)(sizeof(SahBinsOrd) / 4), kBlock), kBlock, 0, st>>>(mb, bins);
    }
    uint32_t m = 1, node_base = 0, levels = 0;
    while (m) {
        if (++levels > 4096 || m > n || node_base >= n) { pt_set_error("pt_build_sah: did not converge"); return PT_ERROR_HIP; }
        for (uint32_t t_lo = 0; t_lo < m; t_lo += kBatch) {
            const uint32_t t_hi = t_lo + kBatch < m ? t_lo + kBatch : m, mb = t_hi - t_lo;
            k_sah_bins<<<nb, kBlock, 0, st>>>(n, ids_a, boxes, tof_a, t_lo, t_hi, tasks_a, bins);
            k_sah_decide<<<mb, 64, 0, st>>>(t_lo, t_hi, tasks_a, bins, max_leaf, boxes, ids_a, inner);
        }
        uint32_t* total_word = count.arm();
        int rc = scan_exclusive(st, mem, inner, offsets, m, total_word, sums);
        if (rc) return rc;
        // launched before the host knows whether any range was split (the flags of a last level are not used):
        // the GPU works on it while the count travels
        k_sah_emit_flags<<<nb, kBlock, 0, st>>>(n, ids_a, boxes, tof_a, tasks_a, flags, m, offsets, node_base, nodes);
        uint32_t n_inner = 0;
        if (count.wait(&n_inner) != hipSuccess) {
            pt_set_error("pt_build_sah: %s", hipGetErrorString(hipGetLastError())); return PT_ERROR_HIP;
        }
        if (n_inner) {
            rc = scan_exclusive(st, mem, flags, left, n, nullptr, sums);
            if (rc) return rc;
            k_sah_scatter<<<nb, kBlock, 0, st>>>(n, ids_a, tof_a, tasks_a, offsets, flags, left, ids_b, tof_b, tasks_b);
            std::swap(ids_a, ids_b); std::swap(tof_a, tof_b); std::swap(tasks_a, tasks_b);
        }
        node_base += n_inner; m = 2 * n_inner;
    }
    // the scratch of this build goes back to the arena here; what comes next on the stream may reuse it, in order
    PT_HIP_CHECK(hipMemcpyAsync(*order_out, ids_a, sizeof(uint32_t) * n, hipMemcpyDeviceToDevice, st));
    PT_HIP_CHECK(hipGetLastError());
    *node_count_out = node_base;
    return PT_OK;
}

int pt_build_wide(PtContext* ctx, const PtBvhNode* nodes_dev, uint32_t node_count, void** wide_out, uint32_t* count_out, uint32_t* depth_out)
{
    hipStream_t st = ctx->stream;
    DevMem mem(ctx);
    uint32_t *front_a = nullptr, *front_b = nullptr, *counts = nullptr, *offsets = nullptr, *sums = nullptr;
    PtWide4* wide = nullptr;
    // every wide node stands for a distinct binary node
    PT_HIP_CHECK(mem.alloc(&front_a, node_count)); PT_HIP_CHECK(mem.alloc(&front_b, node_count));
    PT_HIP_CHECK(mem.alloc(&counts, node_count)); PT_HIP_CHECK(mem.alloc(&offsets, node_count)); PT_HIP_CHECK(mem.alloc(&sums, div_up(node_count, kBlock) + 1));
    uint32_t* total_dev = nullptr;
    PT_HIP_CHECK(mem.alloc(&total_dev, 4));
    LevelCount count(ctx, 1, total_dev);
    const bool own = *wide_out == nullptr;           // otherwise the caller's buffer of node_count nodes
    if (own) PT_HIP_CHECK(hipMalloc((void**)&wide, sizeof(PtWide4) * (size_t)node_count));
    else wide = (PtWide4*)*wide_out;
    auto fail = [&](int code) { if (own) (void)hipFree(wide); return code; };
    if (hipMemsetAsync(front_a, 0, 4, st) != hipSuccess) return fail(PT_ERROR_HIP);   // the root
    uint32_t n_front = 1, level_base = 0, depth = 0;
    while (n_front) {
        ++depth;
        if (level_base + n_front > node_count || depth > 4096) { pt_set_error("pt_build_wide: the binary tree is not a tree"); return fail(PT_ERROR_HIP); }
        const uint32_t nb = div_up(n_front, kBlock);
        uint32_t* total_word = count.arm();
        k_wide_count<<<nb, kBlock, 0, st>>>(nodes_dev, front_a, n_front, counts);
        int rc = scan_exclusive(st, mem, counts, offsets, n_front, total_word, sums);
        if (rc) return fail(rc);
        k_wide_emit<<<nb, kBlock, 0, st>>>(nodes_dev, front_a, n_front, offsets, level_base, wide, front_b);
        uint32_t next = 0;
        if (count.wait(&next) != hipSuccess) {
            pt_set_error("pt_build_wide: %s", hipGetErrorString(hipGetLastError())); return fail(PT_ERROR_HIP);
        }
        level_base += n_front; n_front = next;
        std::swap(front_a, front_b);
    }
    *wide_out = wide; *count_out = level_base; *depth_out = depth;
    return PT_OK;
}

int pt_build_lbvh(PtContext* ctx, const float* boxes, uint32_t n, uint32_t max_leaf,
                  PtBvhNode** nodes_out, uint32_t* node_count_out, uint32_t** leaf_prims_out, uint64_t** sorted_keys_out)
{
    hipStream_t st = ctx->stream;
    DevMem mem(ctx);
    uint32_t* cb = nullptr; uint64_t* keys = nullptr; uint32_t* vals = nullptr;
    PT_HIP_CHECK(mem.alloc(&cb, 8));
    PT_HIP_CHECK(mem.alloc(&keys, n));
    PT_HIP_CHECK(mem.alloc(&vals, n));
    PT_HIP_CHECK(hipMemsetAsync(cb, 0xFF, 12, st));
    PT_HIP_CHECK(hipMemsetAsync(cb + 3, 0x00, 12, st));
    uint32_t nb = div_up(n, kBlock);
    k_centroid_bounds<<<nb, kBlock, 0, st>>>(boxes, n, cb);
    k_morton<<<nb, kBlock, 0, st>>>(boxes, n, cb, keys, vals);
    PT_HIP_CHECK(hipGetLastError());
    int rc = radix_sort_pairs(st, mem, &keys, &vals, n, 64);
    if (rc) return rc;

    // results that outlive this call
    uint32_t* leaf_prims = nullptr; uint64_t* sorted_keys = nullptr; PtBvhNode* nodes = nullptr;
    PT_HIP_CHECK(hipMalloc((void**)&leaf_prims, sizeof(uint32_t) * n));
    PT_HIP_CHECK(hipMalloc((void**)&sorted_keys, sizeof(uint64_t) * n));
    PT_HIP_CHECK(hipMemcpyAsync(leaf_prims, vals, sizeof(uint32_t) * n, hipMemcpyDeviceToDevice, st));
    PT_HIP_CHECK(hipMemcpyAsync(sorted_keys, keys, sizeof(uint64_t) * n, hipMemcpyDeviceToDevice, st));
    auto fail = [&](int code) { (void)hipFree(leaf_prims); (void)hipFree(sorted_keys); if (nodes) (void)hipFree(nodes); return code; };

    if (n == 1) {
        float b[6];
        if (hipMemcpyAsync(b, boxes, 24, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            pt_set_error("pt_build_lbvh: copy of the single box failed"); return fail(PT_ERROR_HIP);
        }
        PtBvhNode nd; memset(&nd, 0, sizeof nd);
        const float inf = std::numeric_limits<float>::infinity();
        for (int k = 0; k < 3; ++k) { nd.lo0[k] = b[k]; nd.hi0[k] = b[3 + k]; nd.lo1[k] = inf; nd.hi1[k] = -inf; }
        nd.c0 = ~0; nd.c1 = std::numeric_limits<int32_t>::min();
        if (hipMalloc((void**)&nodes, sizeof nd) != hipSuccess || hipMemcpy(nodes, &nd, sizeof nd, hipMemcpyHostToDevice) != hipSuccess) {
            pt_set_error("pt_build_lbvh: node upload failed"); return fail(PT_ERROR_HIP);
        }
        *nodes_out = nodes; *node_count_out = 1; *leaf_prims_out = leaf_prims; *sorted_keys_out = sorted_keys;
        return PT_OK;
    }

    const int ni = (int)n - 1;
    int *left = nullptr, *right = nullptr, *parent = nullptr; uint32_t* rng_lo = nullptr;
    InnerData* data = nullptr; uint8_t* collapsed = nullptr; uint32_t* done = nullptr; uint32_t* n_done = nullptr;
    uint32_t* alive = nullptr; uint32_t* new_index = nullptr; uint32_t* total = nullptr;
    hipError_t e = hipSuccess;
    if ((e = mem.alloc(&left, ni)) || (e = mem.alloc(&right, ni)) || (e = mem.alloc(&parent, ni)) || (e = mem.alloc(&rng_lo, ni)) ||
        (e = mem.alloc(&data, ni)) || (e = mem.alloc(&collapsed, ni)) || (e = mem.alloc(&done, ni)) || (e = mem.alloc(&n_done, 4)) ||
        (e = mem.alloc(&alive, ni)) || (e = mem.alloc(&new_index, ni)) || (e = mem.alloc(&total, 4))) {
        pt_set_error("pt_build_lbvh: hipMalloc failed: %s", hipGetErrorString(e)); return fail(PT_ERROR_OUT_OF_MEMORY);
    }
    uint32_t nbi = div_up((uint32_t)ni, kBlock);
    (void)hipMemsetAsync(done, 0, sizeof(uint32_t) * ni, st);
    (void)hipMemsetAsync(n_done, 0, 16, st);
    k_karras<<<nbi, kBlock, 0, st>>>(keys, (int)n, left, right, rng_lo, parent);
    // bottom-up: one launch per tree level; a kernel boundary is the only hand-off between levels
    uint32_t pass = 1, finished = 0;
    while (finished < (uint32_t)ni) {
        for (int k = 0; k < 8; ++k, ++pass)
            k_refit_pass<<<nbi, kBlock, 0, st>>>(ni, pass, left, right, boxes, vals, data, collapsed, done, max_leaf, n_done);
        if (read_word(ctx, n_done, &finished) != hipSuccess) {
            pt_set_error("pt_build_lbvh: refit failed: %s", hipGetErrorString(hipGetLastError())); return fail(PT_ERROR_HIP);
        }
        if (pass > 4096) { pt_set_error("pt_build_lbvh: refit did not converge"); return fail(PT_ERROR_HIP); }
    }
    k_alive<<<nbi, kBlock, 0, st>>>(ni, parent, data, collapsed, max_leaf, alive);
    rc = scan_exclusive(st, mem, alive, new_index, (uint32_t)ni, total);
    if (rc) return fail(rc);
    uint32_t n_alive = 0;
    if (read_word(ctx, total, &n_alive) != hipSuccess) {
        pt_set_error("pt_build_lbvh: scan failed: %s", hipGetErrorString(hipGetLastError())); return fail(PT_ERROR_HIP);
    }
    if (hipMalloc((void**)&nodes, sizeof(PtBvhNode) * (size_t)n_alive) != hipSuccess) { pt_set_error("pt_build_lbvh: out of memory"); return fail(PT_ERROR_OUT_OF_MEMORY); }
    k_emit<<<nbi, kBlock, 0, st>>>(ni, left, right, rng_lo, boxes, vals, data, collapsed, alive, new_index, nodes);
    if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) {
        pt_set_error("pt_build_lbvh: emit failed"); return fail(PT_ERROR_HIP);
    }
    *nodes_out = nodes; *node_count_out = n_alive; *leaf_prims_out = leaf_prims; *sorted_keys_out = sorted_keys;
    return PT_OK;
}

// One cleanup path for the build entry points: the timing events, and the structure under
// construction unless the build reaches its end and releases it.
struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    hipError_t create() { hipError_t e = hipEventCreate(&a); return e != hipSuccess ? e : hipEventCreate(&b); }
    ~EventPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
};
static void blas_free(PtBlas* b)
{
    (void)hipFree(b->nodes); (void)hipFree(b->leaf_prims); (void)hipFree(b->sorted_keys);
    if (b->block) (void)hipFree(b->block);       // tris, wide, trav_nodes and trav_order are parts of it
    else { (void)hipFree(b->tris); (void)hipFree(b->wide); (void)hipFree(b->trav_nodes); (void)hipFree(b->trav_order); }
    delete b;
}
static void tlas_free(PtTlas* t)      // nothing in flight may still traverse the pool
{
    (void)hipFree(t->nodes); (void)hipFree(t->leaf_inst); (void)hipFree(t->inst); (void)hipFree(t->shade_inst); (void)hipFree(t->trav_nodes);
    for (auto& r : t->tri_records) (void)hipFree(r.dev);
    (void)hipFree(t->pool_nodes); (void)hipFree(t->pool_nodes_f32); (void)hipFree(t->pool_tris);
    delete t;
}
template <typename T, void (*FREE)(T*)>
struct Owner {
    T* p;
    explicit Owner(T* q) : p(q) {}
    ~Owner() { if (p) FREE(p); }
    T* release() { T* q = p; p = nullptr; return q; }
};

// =============================================================================================
extern "C" {

// Padded primitive boxes of a BLAS (6 floats each) and scal: [0] bits of the largest |coordinate|, [1..6] the root
// box as ordered uints.
static int blas_boxes(PtContext* ctx, DevMem& mem, const uint8_t* v, uint32_t vertex_stride, const uint32_t* indices_dev,
                      uint32_t prim_count, uint32_t** scal_out, float** boxes_out)
{
    hipStream_t st = ctx->stream;
    uint32_t* scal = nullptr; float* boxes = nullptr;
    PT_HIP_CHECK(mem.alloc(&scal, 8));
    PT_HIP_CHECK(mem.alloc(&boxes, (size_t)prim_count * 6));
    PT_HIP_CHECK(hipMemsetAsync(scal, 0, 4, st));
    PT_HIP_CHECK(hipMemsetAsync(scal + 1, 0xFF, 12, st));
    PT_HIP_CHECK(hipMemsetAsync(scal + 4, 0x00, 12, st));
    const uint32_t nb = div_up(prim_count, kBlock);
    k_blas_maxabs<<<nb, kBlock, 0, st>>>(v, vertex_stride, indices_dev, prim_count, scal);
    k_tri_boxes<<<nb, kBlock, 0, st>>>(v, vertex_stride, indices_dev, prim_count, scal, boxes, scal + 1);
    PT_HIP_CHECK(hipGetLastError());
    *scal_out = scal; *boxes_out = boxes;
    return PT_OK;
}

// The Morton-order tree of a BLAS (what pt_blas_download returns).  The kernels traverse the binned-SAH tree, so
// this one is built when it is first asked for; the vertex and index buffers given to pt_blas_build must still
// hold the same data then (they must anyway: shading reads them).
static int blas_ensure_morton(PtBlas* b)
{
    if (b->nodes) return PT_OK;
    PtContext* ctx = b->ctx;
    PT_HIP_CHECK(hipSetDevice(ctx->device));
    DevMem mem(ctx);
    uint32_t* scal = nullptr; float* boxes = nullptr;
    int rc = blas_boxes(ctx, mem, b->vertices, b->stride, b->indices, b->prim_count, &scal, &boxes);
    if (rc) return rc;
    return pt_build_lbvh(ctx, boxes, b->prim_count, b->max_leaf, &b->nodes, &b->node_count, &b->leaf_prims, &b->sorted_keys);
}

int pt_blas_build(PtContext* ctx, const void* vertices_dev, uint32_t vertex_stride, uint32_t max_vertex,
                  const uint32_t* indices_dev, uint32_t prim_count, uint32_t max_leaf, PtBlas** out)
{
    PT_REQUIRE(ctx && vertices_dev && indices_dev && out, "null argument");
    PT_REQUIRE(prim_count > 0, "prim_count must be positive");
    PT_REQUIRE(prim_count < (1u << 28), "prim_count must be below 2^28");
    PT_REQUIRE(vertex_stride >= 12 && vertex_stride % 4 == 0, "vertex_stride must be a multiple of 4 and at least 12");
    if (max_leaf == 0) max_leaf = 4;
    PT_REQUIRE(max_leaf <= 8, "max_leaf must be at most 8");
    PT_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    EventPair ev;
    PT_HIP_CHECK(ev.create());
    const hipEvent_t ev0 = ev.a, ev1 = ev.b;
    PT_HIP_CHECK(hipEventRecord(ev0, st));
    DevMem mem(ctx);
    uint32_t* scal = nullptr; float* boxes = nullptr;
    const uint32_t nb = div_up(prim_count, kBlock);
    const uint8_t* v = (const uint8_t*)vertices_dev;
    int rc = blas_boxes(ctx, mem, v, vertex_stride, indices_dev, prim_count, &scal, &boxes);
    if (rc) return rc;
    PtBlas* b = new PtBlas();
    memset(b, 0, sizeof *b);
    Owner<PtBlas, blas_free> owner(b);
    b->ctx = ctx; b->vertices = v; b->stride = vertex_stride; b->max_vertex = max_vertex;
    b->indices = indices_dev; b->prim_count = prim_count; b->max_leaf = max_leaf;
    // The kernels traverse a binned-SAH tree (16-18 % fewer box tests than the Morton tree on the dragon
    // meshes), and the Morton tree is left to blas_ensure_morton; PT_HIP_TRAVERSAL_TREE=lbvh (and a BLAS of
    // one primitive) uses the Morton tree for both purposes.
    const char* tree_env = getenv("PT_HIP_TRAVERSAL_TREE");
    const uint32_t* trav_order = nullptr;            // leaf order of the tree the kernels traverse
    if (!(tree_env && strcmp(tree_env, "lbvh") == 0) && prim_count > 1) {
        // one allocation for what outlives the build (an allocation costs as much as twenty of the kernels):
        // binary nodes (at most n - 1), wide nodes (at most as many), triangle records, leaf order
        const size_t n = prim_count;
        if (hipMalloc(&b->block, n * (sizeof(PtBvhNode) + sizeof(PtWide4) + 3 * sizeof(float4) + sizeof(uint32_t))) != hipSuccess) {
            pt_set_error("pt_blas_build: out of memory"); return PT_ERROR_OUT_OF_MEMORY;
        }
        b->trav_nodes = (PtBvhNode*)b->block;
        b->wide = (char*)b->block + n * sizeof(PtBvhNode);
        b->tris = (float4*)((char*)b->wide + n * sizeof(PtWide4));
        b->trav_order = (uint32_t*)((char*)b->tris + n * 3 * sizeof(float4));
        rc = pt_build_sah(ctx, boxes, prim_count, max_leaf, scal + 1, &b->trav_nodes, &b->trav_node_count, &b->trav_order);
        if (rc) return rc;
        trav_order = b->trav_order;
        rc = pt_build_wide(ctx, b->trav_nodes, b->trav_node_count, &b->wide, &b->wide_count, &b->wide_depth);
    } else {
        rc = pt_build_lbvh(ctx, boxes, prim_count, max_leaf, &b->nodes, &b->node_count, &b->leaf_prims, &b->sorted_keys);
        if (rc) return rc;
        trav_order = b->leaf_prims;
        rc = pt_build_wide(ctx, b->nodes, b->node_count, &b->wide, &b->wide_count, &b->wide_depth);
    }
    if (rc) return rc;
    uint32_t host_scal[7];
    hipError_t e = b->tris ? hipSuccess : hipMalloc((void**)&b->tris, sizeof(float4) * 3 * (size_t)prim_count);
    if (e == hipSuccess) {
        k_pack_tris<<<nb, kBlock, 0, st>>>(v, vertex_stride, indices_dev, trav_order, prim_count, b->tris);
        e = hipMemcpyAsync(host_scal, scal, sizeof host_scal, hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipEventRecord(ev1, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { pt_set_error("pt_blas_build: %s", hipGetErrorString(e)); return PT_ERROR_HIP; }
    float maxabs; memcpy(&maxabs, &host_scal[0], 4);
    b->delta = maxabs * 3.0517578125e-05f;
    for (int k = 0; k < 3; ++k) { b->root_lo[k] = ord2f(host_scal[1 + k]); b->root_hi[k] = ord2f(host_scal[4 + k]); }
    (void)hipEventElapsedTime(&b->build_ms, ev0, ev1);
    *out = owner.release();
    return PT_OK;
}

int pt_blas_destroy(PtBlas* b)
{
    if (!b) return PT_OK;
    blas_free(b);
    return PT_OK;
}

int pt_blas_info(const PtBlas* b, PtAsInfo* out)
{
    PT_REQUIRE(b && out, "null argument");
    out->node_count = b->node_count; out->prim_count = b->prim_count;
    for (int k = 0; k < 3; ++k) { out->root_lo[k] = b->root_lo[k]; out->root_hi[k] = b->root_hi[k]; }
    out->delta = b->delta; out->build_ms = b->build_ms;
    return PT_OK;
}

int pt_blas_wide_download(const PtBlas* b, void* nodes64, uint32_t capacity, uint32_t* count_out, uint32_t* depth_out)
{
    PT_REQUIRE(b, "null argument");
    if (count_out) *count_out = b->wide_count;
    if (depth_out) *depth_out = b->wide_depth;
    if (!nodes64) return PT_OK;
    PT_REQUIRE(capacity >= b->wide_count, "capacity is smaller than the number of wide nodes");
    PT_HIP_CHECK(hipSetDevice(b->ctx->device));
    PT_HIP_CHECK(hipMemcpy(nodes64, b->wide, sizeof(PtWide4) * (size_t)b->wide_count, hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_blas_traversal_download(const PtBlas* b, PtBvhNode* nodes, uint32_t capacity, uint32_t* leaf_prims, uint32_t* count_out)
{
    PT_REQUIRE(b, "null argument");
    const PtBvhNode* src = b->trav_nodes ? b->trav_nodes : b->nodes;
    const uint32_t* order = b->trav_nodes ? b->trav_order : b->leaf_prims;
    const uint32_t count = b->trav_nodes ? b->trav_node_count : b->node_count;
    if (count_out) *count_out = count;
    PT_HIP_CHECK(hipSetDevice(b->ctx->device));
    if (nodes) {
        PT_REQUIRE(capacity >= count, "capacity is smaller than the number of nodes");
        PT_HIP_CHECK(hipMemcpy(nodes, src, sizeof(PtBvhNode) * (size_t)count, hipMemcpyDeviceToHost));
    }
    if (leaf_prims) PT_HIP_CHECK(hipMemcpy(leaf_prims, order, sizeof(uint32_t) * b->prim_count, hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_tlas_traversal_download(const PtTlas* t, PtBvhNode* nodes, uint32_t capacity, uint32_t* leaf_instances, uint32_t* count_out)
{
    PT_REQUIRE(t, "null argument");
    const uint32_t count = t->trav_nodes ? t->trav_node_count : t->node_count;
    if (count_out) *count_out = count;
    PT_HIP_CHECK(hipSetDevice(t->ctx->device));
    if (nodes) {
        PT_REQUIRE(capacity >= count, "capacity is smaller than the number of nodes");
        PT_HIP_CHECK(hipMemcpy(nodes, t->trav_nodes ? t->trav_nodes : t->nodes, sizeof(PtBvhNode) * (size_t)count, hipMemcpyDeviceToHost));
    }
    if (leaf_instances) {
        if (t->trav_nodes) memcpy(leaf_instances, t->trav_order.data(), sizeof(uint32_t) * t->n);
        else PT_HIP_CHECK(hipMemcpy(leaf_instances, t->leaf_inst, sizeof(uint32_t) * t->n, hipMemcpyDeviceToHost));
    }
    return PT_OK;
}

int pt_tlas_wide_download(const PtTlas* t, void* nodes64, uint32_t capacity, uint32_t* count_out, uint32_t* depth_out)
{
    PT_REQUIRE(t, "null argument");
    if (count_out) *count_out = t->wide_count;
    if (depth_out) *depth_out = t->wide_depth;
    if (!nodes64) return PT_OK;
    PT_REQUIRE(capacity >= t->wide_count, "capacity is smaller than the number of wide nodes");
    PT_HIP_CHECK(hipSetDevice(t->ctx->device));
    PT_HIP_CHECK(hipMemcpy(nodes64, t->pool_nodes, sizeof(PtWide4) * (size_t)t->wide_count, hipMemcpyDeviceToHost));
    // the pool marks unused slots its own way (k_wide_rebase); what is handed out is the builder's form
    PtWide4* w = (PtWide4*)nodes64;
    for (uint32_t i = 0; i < t->wide_count; ++i)
        for (int e = 0; e < 4; ++e)
            if (pt_wide_slot_unused(w[i], e)) {
                w[i].child[e] = PT_WIDE_EMPTY;
                for (int k = 0; k < 6; ++k) w[i].q[k] &= ~(255u << (8 * e));
            }
    return PT_OK;
}

int pt_blas_download(const PtBlas* b, PtBvhNode* nodes, uint32_t* leaf_prims, uint64_t* sorted_keys)
{
    PT_REQUIRE(b, "null argument");
    if (int rc = blas_ensure_morton(const_cast<PtBlas*>(b))) return rc;
    PT_HIP_CHECK(hipSetDevice(b->ctx->device));
    if (nodes) PT_HIP_CHECK(hipMemcpy(nodes, b->nodes, sizeof(PtBvhNode) * b->node_count, hipMemcpyDeviceToHost));
    if (leaf_prims) PT_HIP_CHECK(hipMemcpy(leaf_prims, b->leaf_prims, sizeof(uint32_t) * b->prim_count, hipMemcpyDeviceToHost));
    if (sorted_keys) PT_HIP_CHECK(hipMemcpy(sorted_keys, b->sorted_keys, sizeof(uint64_t) * b->prim_count, hipMemcpyDeviceToHost));
    return PT_OK;
}

// Inverse of a row-major 3x4 affine transform in binary64 by cofactors, rounded once to binary32:
// the world->object matrix the driver would expose as gl_WorldToObjectEXT.
static void invert_transform(const float m[12], float w[12])
{
    double a00 = m[0], a01 = m[1], a02 = m[2], t0 = m[3];
    double a10 = m[4], a11 = m[5], a12 = m[6], t1 = m[7];
    double a20 = m[8], a21 = m[9], a22 = m[10], t2 = m[11];
    double c00 = a11 * a22 - a12 * a21;
    double c01 = a12 * a20 - a10 * a22;
    double c02 = a10 * a21 - a11 * a20;
    double det = (a00 * c00 + a01 * c01) + a02 * c02;
    double i00 = c00 / det, i01 = (a02 * a21 - a01 * a22) / det, i02 = (a01 * a12 - a02 * a11) / det;
    double i10 = c01 / det, i11 = (a00 * a22 - a02 * a20) / det, i12 = (a02 * a10 - a00 * a12) / det;
    double i20 = c02 / det, i21 = (a01 * a20 - a00 * a21) / det, i22 = (a00 * a11 - a01 * a10) / det;
    w[0] = (float)i00; w[1] = (float)i01; w[2] = (float)i02;  w[3] = (float)(-((i00 * t0 + i01 * t1) + i02 * t2));
    w[4] = (float)i10; w[5] = (float)i11; w[6] = (float)i12;  w[7] = (float)(-((i10 * t0 + i11 * t1) + i12 * t2));
    w[8] = (float)i20; w[9] = (float)i21; w[10] = (float)i22; w[11] = (float)(-((i20 * t0 + i21 * t1) + i22 * t2));
}

// Largest TLAS whose rays start with the instances instead of the root node.  Measured on one GPU: two instances
// (dragon scene) +4.9 % on the frame, eight (Cornell box) -27 %, six (helmet) -9 %: a missed instance costs a pass
// of the instance phase where the root step would have culled it, so the root step pays from about three on.
#ifndef PT_ROOT_ITEMS_DEFAULT
#define PT_ROOT_ITEMS_DEFAULT 2u
#endif
#ifndef PT_TINY_BLAS
#define PT_TINY_BLAS 4u
#endif

int pt_tlas_build(PtContext* ctx, const PtInstance* instances, uint32_t n, PtTlas** out)
{
    PT_REQUIRE(ctx && instances && out, "null argument");
    PT_REQUIRE(n > 0, "at least one instance is required");
    PT_REQUIRE(n <= (1u << 24), "instance count exceeds the 24-bit custom index");
    PT_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    EventPair ev;
    PT_HIP_CHECK(ev.create());
    const hipEvent_t ev0 = ev.a, ev1 = ev.b;
    PT_HIP_CHECK(hipEventRecord(ev0, st));
    PtTlas* t = new PtTlas();
    Owner<PtTlas, tlas_free> owner(t);        // nothing traverses a structure that is still being built
    t->ctx = ctx; t->n = n; t->nodes = nullptr; t->leaf_inst = nullptr; t->inst = nullptr; t->shade_inst = nullptr;
    t->pool_nodes = nullptr; t->pool_nodes_f32 = nullptr; t->pool_tris = nullptr; t->pool_node_count = 0; t->n_root = 0;
    t->trav_nodes = nullptr; t->trav_node_count = 0;
    t->host_boxes.resize((size_t)n * 6); t->host_w2o.resize((size_t)n * 12);
    std::vector<PtDevInstance> di(n);
    t->host_o2w.resize((size_t)n * 12); t->host_custom.resize(n); t->host_prim_count.resize(n); t->shade_version = 0;
    const float inf = std::numeric_limits<float>::infinity();
    for (int k = 0; k < 3; ++k) { t->root_lo[k] = inf; t->root_hi[k] = -inf; }
    for (uint32_t i = 0; i < n; ++i) {
        const PtInstance& I = instances[i];
        const PtBlas* b = (const PtBlas*)(uintptr_t)I.accelerationStructureReference;
        if (!b) { pt_set_error("pt_tlas_build: instance %u has a null BLAS reference", i); return PT_ERROR_INVALID_ARGUMENT; }
        const float* m = I.transform;
        float w2o[12];
        invert_transform(m, w2o);
        // world box: the eight corners of the BLAS root box through object->world, then padded
        float lo[3] = { inf, inf, inf }, hi[3] = { -inf, -inf, -inf };
        for (int c = 0; c < 8; ++c) {
            float px = (c & 1) ? b->root_hi[0] : b->root_lo[0];
            float py = (c & 2) ? b->root_hi[1] : b->root_lo[1];
            float pz = (c & 4) ? b->root_hi[2] : b->root_lo[2];
            for (int r = 0; r < 3; ++r) {
                float q = __builtin_fmaf(m[4 * r], px, __builtin_fmaf(m[4 * r + 1], py, __builtin_fmaf(m[4 * r + 2], pz, m[4 * r + 3])));
                lo[r] = q < lo[r] ? q : lo[r];
                hi[r] = q > hi[r] ? q : hi[r];
            }
        }
        float maxabs = 0.0f;
        for (int r = 0; r < 3; ++r) { maxabs = fmaxf(maxabs, fabsf(lo[r])); maxabs = fmaxf(maxabs, fabsf(hi[r])); }
        float dw = maxabs * 3.0517578125e-05f;
        for (int r = 0; r < 3; ++r) {
            lo[r] -= dw; hi[r] += dw;
            t->host_boxes[6 * (size_t)i + r] = lo[r]; t->host_boxes[6 * (size_t)i + 3 + r] = hi[r];
            t->root_lo[r] = fminf(t->root_lo[r], lo[r]); t->root_hi[r] = fmaxf(t->root_hi[r], hi[r]);
        }
        memcpy(&t->host_w2o[12 * (size_t)i], w2o, 48);
        uint32_t custom = I.instanceCustomIndexAndMask & 0xFFFFFFu;
        float customf; memcpy(&customf, &custom, 4);
        for (int r = 0; r < 3; ++r) {
            di[i].q[r] = make_float4(w2o[4 * r], w2o[4 * r + 1], w2o[4 * r + 2], w2o[4 * r + 3]);
        }
        memcpy(&t->host_o2w[12 * (size_t)i], m, 48);
        t->host_custom[i] = custom;
        t->host_prim_count[i] = b->prim_count;
        di[i].q[3] = make_float4(lo[0], lo[1], lo[2], b->delta);
        di[i].q[4] = make_float4(hi[0], hi[1], hi[2], customf);
        di[i].q[5] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // x = bits(root node of the BLAS in the pool), set below
    }
    float* boxes_dev = nullptr;
    hipError_t e = hipMalloc((void**)&boxes_dev, sizeof(float) * 6 * (size_t)n);
    if (e == hipSuccess) e = hipMemcpyAsync(boxes_dev, t->host_boxes.data(), sizeof(float) * 6 * (size_t)n, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) { (void)hipFree(boxes_dev); pt_set_error("pt_tlas_build: %s", hipGetErrorString(e)); return PT_ERROR_HIP; }
    uint64_t* keys = nullptr;
    int rc = pt_build_lbvh(ctx, boxes_dev, n, 1, &t->nodes, &t->node_count, &t->leaf_inst, &keys);
    (void)hipFree(boxes_dev);
    if (rc) return rc;
    (void)hipFree(keys);
    // pool: wide nodes of the TLAS followed by those of every distinct BLAS; triangles of every distinct BLAS
    {
        std::vector<const PtBlas*> uniq;
        std::vector<uint32_t> tri_base;
        std::vector<uint32_t> blas_of_inst(n);
        uint32_t nt = 0;
        for (uint32_t i = 0; i < n; ++i) {
            const PtBlas* b = (const PtBlas*)(uintptr_t)instances[i].accelerationStructureReference;
            size_t k = 0;
            while (k < uniq.size() && uniq[k] != b) ++k;
            if (k == uniq.size()) { uniq.push_back(b); tri_base.push_back(nt); nt += b->prim_count; }
            blas_of_inst[i] = (uint32_t)k;
        }
        if (nt >= (1u << 28)) { pt_set_error("pt_tlas_build: more than 2^28 triangles in one TLAS"); return PT_ERROR_LIMIT; }
        // The layout the kernels traverse is derived from the cheaper (surface-area model) of the Morton tree, which
        // stays what pt_tlas_download returns, and a full-sweep SAH tree over the instance boxes built on the host
        // (instances are few: the tables hold at most 1024).  Measured: composite scene, 15 instances, cost 9.50 -> 8.74,
        // +2.5 % rays/s; helmet 4.35 -> 4.17; Cornell box 6.76 against 6.88, the Morton tree stays.
        // PT_HIP_TLAS_TREE=lbvh|sweep forces one of them (a TLAS of one instance or of more than kSweepMax has the Morton tree only).
        constexpr uint32_t kSweepMax = 4096;
        const char* tt_env = getenv("PT_HIP_TLAS_TREE");
        if (n >= 2 && n <= kSweepMax && !(tt_env && strcmp(tt_env, "lbvh") == 0)) {
            std::vector<PtBvhNode> hn, morton(t->node_count);
            pt_sweep_sah_build_host(t->host_boxes.data(), n, hn, t->trav_order);
            e = hipStreamSynchronize(st);        // the Morton tree was built on the context's stream
            if (e == hipSuccess) e = hipMemcpy(morton.data(), t->nodes, sizeof(PtBvhNode) * morton.size(), hipMemcpyDeviceToHost);
            const bool forced = tt_env && strcmp(tt_env, "sweep") == 0;
            if (e == hipSuccess && (forced || pt_tree_cost_host(hn.data(), hn.size()) < pt_tree_cost_host(morton.data(), morton.size()))) {
                e = hipMalloc((void**)&t->trav_nodes, sizeof(PtBvhNode) * hn.size());
                if (e == hipSuccess) e = hipMemcpy(t->trav_nodes, hn.data(), sizeof(PtBvhNode) * hn.size(), hipMemcpyHostToDevice);
                t->trav_node_count = (uint32_t)hn.size();
            }
            if (e != hipSuccess) { pt_set_error("pt_tlas_build: %s", hipGetErrorString(e)); return PT_ERROR_HIP; }
        }
        void* tlas_wide = nullptr;
        rc = t->trav_nodes ? pt_build_wide(ctx, t->trav_nodes, t->trav_node_count, &tlas_wide, &t->wide_count, &t->wide_depth)
                           : pt_build_wide(ctx, t->nodes, t->node_count, &tlas_wide, &t->wide_count, &t->wide_depth);
        if (rc) return rc;
        uint64_t total_nodes64 = t->wide_count;
        uint32_t depth_blas = 0;
        std::vector<uint32_t> root_of(uniq.size());
        for (size_t k = 0; k < uniq.size(); ++k) {
            root_of[k] = (uint32_t)total_nodes64; total_nodes64 += uniq[k]->wide_count;
            depth_blas = uniq[k]->wide_depth > depth_blas ? uniq[k]->wide_depth : depth_blas;
            // the float-plane step addresses 128-byte nodes with a 32-bit byte offset (pt_wide_step<true>)
            if (total_nodes64 >= PT_MAX_POOL_NODES) {
                pt_set_error("pt_tlas_build: the traversal pool would hold %llu wide nodes, the limit is %u", (unsigned long long)total_nodes64, PT_MAX_POOL_NODES - 1u);
                (void)hipFree(tlas_wide); return PT_ERROR_LIMIT;
            }
        }
        const uint32_t total_nodes = (uint32_t)total_nodes64;
        // a step pushes at most three references and an instance entry one
        // (or, for a ray that starts with the instances of a small TLAS, all but one of them)
        const uint32_t tlas_part = std::max(3 * t->wide_depth, n <= PT_ROOT_ITEMS_MAX ? n - 1 : 0u);
        if (tlas_part + 1 + 3 * depth_blas > PT_MIN_LDS_STACK + PT_SPILL_STACK) {      // the shallowest LDS part of any traversal kernel
            pt_set_error("pt_tlas_build: acceleration structure too deep for the traversal stack (%u + %u wide levels)", t->wide_depth, depth_blas);
            (void)hipFree(tlas_wide); return PT_ERROR_LIMIT;
        }
        // What a ray that enters instance i continues with: the BLAS's root node, or, for a BLAS of
        // at most PT_TINY_BLAS triangles, one leaf over all of them (the instance's own box test has
        // just passed; a node step in front of so few triangles costs more than it rejects).
        const char* tiny_env = getenv("PT_HIP_TINY_BLAS");
        const uint32_t tiny = tiny_env ? (uint32_t)atoi(tiny_env) : PT_TINY_BLAS;
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t k = blas_of_inst[i];
            int32_t entry = (int32_t)root_of[k];
            if (uniq[k]->prim_count <= tiny && uniq[k]->prim_count <= 8u)
                entry = (int32_t)~((tri_base[k] << 3) | (uniq[k]->prim_count - 1u));
            float f; memcpy(&f, &entry, 4); di[i].q[5].x = f;
        }
        e = hipMalloc((void**)&t->pool_nodes, sizeof(PtWide4) * (size_t)total_nodes);
        if (e == hipSuccess) e = hipMalloc((void**)&t->pool_tris, sizeof(float4) * 3 * (size_t)(nt ? nt : 1));
        if (e == hipSuccess) {
            k_wide_rebase<<<div_up(t->wide_count, kBlock), kBlock, 0, st>>>((const PtWide4*)tlas_wide, t->wide_count, 0, 0, (PtWide4*)t->pool_nodes);
            for (size_t k = 0; k < uniq.size(); ++k)
                k_wide_rebase<<<div_up(uniq[k]->wide_count, kBlock), kBlock, 0, st>>>((const PtWide4*)uniq[k]->wide, uniq[k]->wide_count, root_of[k], tri_base[k], (PtWide4*)t->pool_nodes);
            e = hipGetLastError();
        }
        if (e == hipSuccess) {       // the same nodes with float planes, for the kernel whose waves walk them in step
            e = hipMalloc(&t->pool_nodes_f32, sizeof(PtWideF) * (size_t)total_nodes);
            if (e == hipSuccess) k_wide_expand<<<div_up(total_nodes, kBlock), kBlock, 0, st>>>((const PtWide4*)t->pool_nodes, total_nodes, (PtWideF*)t->pool_nodes_f32);
        }
        for (size_t k = 0; k < uniq.size() && e == hipSuccess; ++k)
            e = hipMemcpyAsync(t->pool_tris + 3 * (size_t)tri_base[k], uniq[k]->tris, sizeof(float4) * 3 * (size_t)uniq[k]->prim_count, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        (void)hipFree(tlas_wide);
        if (e != hipSuccess) { pt_set_error("pt_tlas_build: %s", hipGetErrorString(e)); return PT_ERROR_HIP; }
        t->pool_node_count = total_nodes;
        // over a few instances rays start with the instances themselves (pt_ray_start); PT_HIP_ROOT_ITEMS sets the
        // largest such TLAS (0: every ray starts at the root node)
        const char* ri = getenv("PT_HIP_ROOT_ITEMS");
        const uint32_t most = ri ? (uint32_t)atoi(ri) : PT_ROOT_ITEMS_DEFAULT;
        t->n_root = (n <= most && n <= PT_ROOT_ITEMS_MAX) ? (int)n : 0;
    }
    // instance records in the order of the TLAS leaves, so a leaf reference addresses its record
    // directly; the instance's own index (the tie-break key and what a hit reports) sits in q[5].y
    std::vector<uint32_t> slot_inst(n);
    std::vector<PtDevInstance> di_sorted(n);
    if (t->trav_nodes) { slot_inst = t->trav_order; e = hipSuccess; }
    else e = hipMemcpy(slot_inst.data(), t->leaf_inst, sizeof(uint32_t) * n, hipMemcpyDeviceToHost);
    if (e == hipSuccess) {
        for (uint32_t k = 0; k < n; ++k) {
            di_sorted[k] = di[slot_inst[k]];
            float f; memcpy(&f, &slot_inst[k], 4); di_sorted[k].q[5].y = f;
        }
        e = hipMalloc((void**)&t->inst, sizeof(PtDevInstance) * n);
    }
    if (e == hipSuccess) e = hipMalloc((void**)&t->shade_inst, sizeof(PtDevShadeInstance) * n);
    if (e == hipSuccess) e = hipMemcpyAsync(t->inst, di_sorted.data(), sizeof(PtDevInstance) * n, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipEventRecord(ev1, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { pt_set_error("pt_tlas_build: %s", hipGetErrorString(e)); return PT_ERROR_HIP; }
    (void)hipEventElapsedTime(&t->build_ms, ev0, ev1);
    *out = owner.release();
    return PT_OK;
}

int pt_tlas_destroy(PtTlas* t)
{
    if (!t) return PT_OK;
    (void)pt_lanes_sync(t->ctx);   // frames in flight traverse the pool
    tlas_free(t);
    return PT_OK;
}

int pt_tlas_info(const PtTlas* t, PtAsInfo* out)
{
    PT_REQUIRE(t && out, "null argument");
    out->node_count = t->node_count; out->prim_count = t->n;
    for (int k = 0; k < 3; ++k) { out->root_lo[k] = t->root_lo[k]; out->root_hi[k] = t->root_hi[k]; }
    out->delta = 0.0f; out->build_ms = t->build_ms;
    return PT_OK;
}

int pt_tlas_download(const PtTlas* t, PtBvhNode* nodes, uint32_t* leaf_instances, float* instance_boxes, float* instance_w2o)
{
    PT_REQUIRE(t, "null argument");
    PT_HIP_CHECK(hipSetDevice(t->ctx->device));
    if (nodes) PT_HIP_CHECK(hipMemcpy(nodes, t->nodes, sizeof(PtBvhNode) * t->node_count, hipMemcpyDeviceToHost));
    if (leaf_instances) PT_HIP_CHECK(hipMemcpy(leaf_instances, t->leaf_inst, sizeof(uint32_t) * t->n, hipMemcpyDeviceToHost));
    if (instance_boxes) memcpy(instance_boxes, t->host_boxes.data(), sizeof(float) * 6 * (size_t)t->n);
    if (instance_w2o) memcpy(instance_w2o, t->host_w2o.data(), sizeof(float) * 12 * (size_t)t->n);
    return PT_OK;
}

} // extern "C"
