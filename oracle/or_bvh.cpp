/* or_bvh.cpp -- canonical LBVH + SAH leaf-collapse builder of the CPU oracle
 * (test infrastructure only).
 *
 * Stands in for the driver BVH builds behind
 *   source/bottom_level_acceleration_structure.cpp:74-77 (BLAS, ePreferFastTrace, eBuild)
 *   source/top_level_acceleration_structure.cpp:109-112  (TLAS)
 * The reference leaves the structure to the Vulkan driver; this file defines the one the
 * HIP builders must reproduce bit for bit:
 *   1. primitive boxes -> centroids -> 63-bit Morton keys (21 bits per axis)
 *   2. stable sort by key
 *   3. Karras 2012 radix-tree topology with (key, position) tie-break
 *   4. bottom-up boxes, primitive counts and area-weighted SAH cost; a subtree with
 *      <= max_leaf primitives whose leaf cost does not exceed its subtree cost collapses
 *   5. surviving inner nodes are numbered in radix-tree index order and emitted as
 *      64-byte nodes holding both child boxes.
 */
#include "or_internal.h"
#include <algorithm>
#include <limits>

static inline uint64_t expand21(uint32_t v)
{
    uint64_t x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

static inline uint32_t quantize21(float c, float cmin, float cmax)
{
    float ext = cmax - cmin;
    float scale = ext > 0.0f ? 2097152.0f / ext : 0.0f;
    float q = (c - cmin) * scale;
    q = or_min(or_max(q, 0.0f), 2097151.0f);
    return (uint32_t)q;
}

static inline float half_area(const Box& b)
{
    float dx = b.hi.x - b.lo.x, dy = b.hi.y - b.lo.y, dz = b.hi.z - b.lo.z;
    return or_fma(dz, dx, or_fma(dy, dz, dx * dy));
}

static inline Box box_union(const Box& a, const Box& b)
{
    Box r;
    r.lo = V3 { or_min(a.lo.x, b.lo.x), or_min(a.lo.y, b.lo.y), or_min(a.lo.z, b.lo.z) };
    r.hi = V3 { or_max(a.hi.x, b.hi.x), or_max(a.hi.y, b.hi.y), or_max(a.hi.z, b.hi.z) };
    return r;
}

static inline int clz64(uint64_t x) { return x ? __builtin_clzll(x) : 64; }
static inline int clz32(uint32_t x) { return x ? __builtin_clz(x) : 32; }

static inline void store_box(float lo[3], float hi[3], const Box& b)
{
    lo[0] = b.lo.x; lo[1] = b.lo.y; lo[2] = b.lo.z;
    hi[0] = b.hi.x; hi[1] = b.hi.y; hi[2] = b.hi.z;
}

void or_build_lbvh(const std::vector<Box>& boxes, uint32_t max_leaf,
                   std::vector<OrBvhNode>& nodes, std::vector<uint32_t>& leaf_prims,
                   std::vector<uint64_t>* sorted_keys_out)
{
    const uint32_t n = (uint32_t)boxes.size();
    nodes.clear(); leaf_prims.clear();
    if (sorted_keys_out) sorted_keys_out->clear();
    if (n == 0) return;
    const float inf = std::numeric_limits<float>::infinity();

    /* 1. centroids and their bounds */
    std::vector<V3> cen(n);
    V3 cmin { inf, inf, inf }, cmax { -inf, -inf, -inf };
    for (uint32_t i = 0; i < n; ++i) {
        cen[i] = V3 { (boxes[i].lo.x + boxes[i].hi.x) * 0.5f, (boxes[i].lo.y + boxes[i].hi.y) * 0.5f, (boxes[i].lo.z + boxes[i].hi.z) * 0.5f };
        cmin = V3 { or_min(cmin.x, cen[i].x), or_min(cmin.y, cen[i].y), or_min(cmin.z, cen[i].z) };
        cmax = V3 { or_max(cmax.x, cen[i].x), or_max(cmax.y, cen[i].y), or_max(cmax.z, cen[i].z) };
    }
    /* 2. keys, stable sort */
    std::vector<uint64_t> keys(n);
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t qx = quantize21(cen[i].x, cmin.x, cmax.x);
        uint32_t qy = quantize21(cen[i].y, cmin.y, cmax.y);
        uint32_t qz = quantize21(cen[i].z, cmin.z, cmax.z);
        keys[i] = (expand21(qx) << 2) | (expand21(qy) << 1) | expand21(qz);
    }
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });
    std::vector<uint64_t> sk(n);
    for (uint32_t i = 0; i < n; ++i) sk[i] = keys[order[i]];
    leaf_prims = order;
    if (sorted_keys_out) *sorted_keys_out = sk;

    const int32_t EMPTY = std::numeric_limits<int32_t>::min();
    if (n == 1) {
        OrBvhNode nd; memset(&nd, 0, sizeof nd);
        store_box(nd.lo0, nd.hi0, boxes[order[0]]);
        Box e { V3 { inf, inf, inf }, V3 { -inf, -inf, -inf } };
        store_box(nd.lo1, nd.hi1, e);
        nd.c0 = ~(int32_t)0; /* first 0, count 1 */
        nd.c1 = EMPTY;
        nodes.push_back(nd);
        return;
    }

    /* 3. Karras topology.  Inner node i in [0, n-2]; child id >= n-1 ... encoded separately. */
    auto delta = [&](int i, int j) -> int {
        if (j < 0 || j >= (int)n) return -1;
        uint64_t a = sk[i], b = sk[j];
        if (a == b) return 64 + clz32((uint32_t)i ^ (uint32_t)j);
        return clz64(a ^ b);
    };
    const uint32_t ni = n - 1;
    std::vector<int32_t> left(ni), right(ni);       /* >=0 inner, <0 : ~leafpos */
    std::vector<uint32_t> rng_lo(ni), rng_hi(ni);
    std::vector<int32_t> parent_inner(ni, -1), parent_leaf(n, -1);
    for (int i = 0; i < (int)ni; ++i) {
        int d = (delta(i, i + 1) - delta(i, i - 1)) >= 0 ? 1 : -1;
        int dmin = delta(i, i - d);
        int lmax = 2;
        while (delta(i, i + lmax * d) > dmin) lmax *= 2;
        int l = 0;
        for (int t = lmax / 2; t >= 1; t /= 2)
            if (delta(i, i + (l + t) * d) > dmin) l += t;
        int j = i + l * d;
        int dnode = delta(i, j);
        int s = 0;
        for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
            if (delta(i, i + (s + t) * d) > dnode) s += t;
            if (t == 1) break;
        }
        int gamma = i + s * d + std::min(d, 0);
        int lo = std::min(i, j), hi = std::max(i, j);
        rng_lo[i] = lo; rng_hi[i] = hi;
        if (lo == gamma) { left[i] = ~gamma; parent_leaf[gamma] = i; } else { left[i] = gamma; parent_inner[gamma] = i; }
        if (hi == gamma + 1) { right[i] = ~(gamma + 1); parent_leaf[gamma + 1] = i; } else { right[i] = gamma + 1; parent_inner[gamma + 1] = i; }
    }

    /* 4. bottom-up: boxes, counts, cost, collapse flags.  Process by post-order via explicit stack. */
    std::vector<Box> ibox(ni);
    std::vector<uint32_t> icount(ni);
    std::vector<float> icost(ni);
    std::vector<uint8_t> collapsed(ni, 0);
    {
        std::vector<std::pair<int32_t, int>> stack;
        stack.push_back({ 0, 0 });
        while (!stack.empty()) {
            auto& top = stack.back();
            int32_t i = top.first;
            if (top.second == 0) {
                top.second = 1;
                if (left[i] >= 0) stack.push_back({ left[i], 0 });
                if (right[i] >= 0) stack.push_back({ right[i], 0 });
                continue;
            }
            stack.pop_back();
            Box lb, rb; uint32_t lc, rc; float lcost, rcost;
            if (left[i] >= 0) { lb = ibox[left[i]]; lc = icount[left[i]]; lcost = icost[left[i]]; }
            else { lb = boxes[order[~left[i]]]; lc = 1; lcost = half_area(lb); }
            if (right[i] >= 0) { rb = ibox[right[i]]; rc = icount[right[i]]; rcost = icost[right[i]]; }
            else { rb = boxes[order[~right[i]]]; rc = 1; rcost = half_area(rb); }
            ibox[i] = box_union(lb, rb);
            icount[i] = lc + rc;
            float A = half_area(ibox[i]);
            float cint = A + (lcost + rcost);
            float cleaf = (float)icount[i] * A;
            if (i != 0 && icount[i] <= max_leaf && cleaf <= cint) { collapsed[i] = 1; icost[i] = cleaf; }
            else icost[i] = cint;
        }
    }
    /* 5. alive = not collapsed and no collapsed proper ancestor; number in index order */
    std::vector<int32_t> new_index(ni, -1);
    uint32_t alive = 0;
    for (uint32_t i = 0; i < ni; ++i) {
        bool dead = collapsed[i];
        for (int32_t p = parent_inner[i]; !dead && p >= 0 && icount[p] <= max_leaf; p = parent_inner[p])
            if (collapsed[p]) dead = true;
        if (!dead) new_index[i] = (int32_t)alive++;
    }
    nodes.resize(alive);
    for (uint32_t i = 0; i < ni; ++i) {
        if (new_index[i] < 0) continue;
        OrBvhNode nd; memset(&nd, 0, sizeof nd);
        auto child = [&](int32_t c, float lo[3], float hi[3]) -> int32_t {
            if (c < 0) { store_box(lo, hi, boxes[order[~c]]); return ~(int32_t)(((uint32_t)(~c) << 3) | 0u); }
            store_box(lo, hi, ibox[c]);
            if (collapsed[c]) return ~(int32_t)((rng_lo[c] << 3) | (icount[c] - 1));
            return new_index[c];
        };
        nd.c0 = child(left[i], nd.lo0, nd.hi0);
        nd.c1 = child(right[i], nd.lo1, nd.hi1);
        nodes[new_index[i]] = nd;
    }
}

/* ------------------------------------------------------------------------------------------ */
extern "C" {

/* BLAS over prim_count triangles: R32G32B32 positions at vertex_stride, u32 indices
 * (source/renderer.cpp:427-441). */
OrBlas* or_blas_build(const void* vertices, uint32_t vertex_stride, uint32_t max_vertex,
                      const uint32_t* indices, uint32_t prim_count, uint32_t max_leaf)
{
    OrBlas* b = new OrBlas();
    b->vertices = (const uint8_t*)vertices; b->stride = vertex_stride; b->max_vertex = max_vertex;
    b->indices = indices; b->prim_count = prim_count; b->max_leaf = max_leaf;
    const float inf = std::numeric_limits<float>::infinity();
    /* delta = 2^-15 * largest |coordinate| of any referenced vertex */
    float maxabs = 0.0f;
    for (uint32_t p = 0; p < prim_count; ++p) {
        V3 v[3]; b->tri(p, v[0], v[1], v[2]);
        for (int k = 0; k < 3; ++k) {
            maxabs = or_max(maxabs, fabsf(v[k].x)); maxabs = or_max(maxabs, fabsf(v[k].y)); maxabs = or_max(maxabs, fabsf(v[k].z));
        }
    }
    b->delta = maxabs * 3.0517578125e-05f;
    std::vector<Box> boxes(prim_count);
    Box root { V3 { inf, inf, inf }, V3 { -inf, -inf, -inf } };
    for (uint32_t p = 0; p < prim_count; ++p) {
        V3 a, bb, c; b->tri(p, a, bb, c);
        Box x;
        x.lo = V3 { or_min(or_min(a.x, bb.x), c.x) - b->delta, or_min(or_min(a.y, bb.y), c.y) - b->delta, or_min(or_min(a.z, bb.z), c.z) - b->delta };
        x.hi = V3 { or_max(or_max(a.x, bb.x), c.x) + b->delta, or_max(or_max(a.y, bb.y), c.y) + b->delta, or_max(or_max(a.z, bb.z), c.z) + b->delta };
        boxes[p] = x;
        root = box_union(root, x);
    }
    b->root = root;
    or_build_lbvh(boxes, max_leaf, b->nodes, b->leaf_prims, &b->sorted_keys);
    return b;
}
void or_blas_destroy(OrBlas* b) { delete b; }
uint32_t or_blas_node_count(const OrBlas* b) { return (uint32_t)b->nodes.size(); }
const OrBvhNode* or_blas_nodes(const OrBlas* b) { return b->nodes.data(); }
const uint32_t* or_blas_leaf_prims(const OrBlas* b) { return b->leaf_prims.data(); }
const uint64_t* or_blas_sorted_keys(const OrBlas* b) { return b->sorted_keys.data(); }
void or_blas_root_box(const OrBlas* b, float lo[3], float hi[3]) { store_box(lo, hi, b->root); }
float or_blas_delta(const OrBlas* b) { return b->delta; }

/* TLAS over VkAccelerationStructureInstanceKHR-shaped records
 * (source/top_level_acceleration_structure.cpp:21-43). */
OrTlas* or_tlas_build(const OrInstance* instances, uint32_t n)
{
    OrTlas* t = new OrTlas();
    t->inst.resize(n);
    std::vector<Box> boxes(n);
    const float inf = std::numeric_limits<float>::infinity();
    for (uint32_t i = 0; i < n; ++i) {
        OrInst& I = t->inst[i];
        memcpy(I.o2w, instances[i].transform, 48);
        or_invert_transform(I.o2w, I.w2o);
        I.custom_index = instances[i].customIndexAndMask & 0xFFFFFFu;
        I.blas = (const OrBlas*)(uintptr_t)instances[i].blasRef;
        const Box& r = I.blas->root;
        Box w { V3 { inf, inf, inf }, V3 { -inf, -inf, -inf } };
        for (int c = 0; c < 8; ++c) {
            V3 p { (c & 1) ? r.hi.x : r.lo.x, (c & 2) ? r.hi.y : r.lo.y, (c & 4) ? r.hi.z : r.lo.z };
            V3 q = or_xform_point(I.o2w, p);
            w.lo = V3 { or_min(w.lo.x, q.x), or_min(w.lo.y, q.y), or_min(w.lo.z, q.z) };
            w.hi = V3 { or_max(w.hi.x, q.x), or_max(w.hi.y, q.y), or_max(w.hi.z, q.z) };
        }
        float maxabs = 0.0f;
        maxabs = or_max(maxabs, fabsf(w.lo.x)); maxabs = or_max(maxabs, fabsf(w.lo.y)); maxabs = or_max(maxabs, fabsf(w.lo.z));
        maxabs = or_max(maxabs, fabsf(w.hi.x)); maxabs = or_max(maxabs, fabsf(w.hi.y)); maxabs = or_max(maxabs, fabsf(w.hi.z));
        float dw = maxabs * 3.0517578125e-05f;
        I.box.lo = V3 { w.lo.x - dw, w.lo.y - dw, w.lo.z - dw };
        I.box.hi = V3 { w.hi.x + dw, w.hi.y + dw, w.hi.z + dw };
        boxes[i] = I.box;
    }
    or_build_lbvh(boxes, 1, t->nodes, t->leaf_inst, nullptr);
    return t;
}
void or_tlas_destroy(OrTlas* t) { delete t; }
uint32_t or_tlas_node_count(const OrTlas* t) { return (uint32_t)t->nodes.size(); }
const OrBvhNode* or_tlas_nodes(const OrTlas* t) { return t->nodes.data(); }
const uint32_t* or_tlas_leaf_instances(const OrTlas* t) { return t->leaf_inst.data(); }
void or_tlas_instance_box(const OrTlas* t, uint32_t i, float lo[3], float hi[3]) { store_box(lo, hi, t->inst[i].box); }
void or_tlas_instance_w2o(const OrTlas* t, uint32_t i, float w2o[12]) { memcpy(w2o, t->inst[i].w2o, 48); }

} /* extern "C" */
