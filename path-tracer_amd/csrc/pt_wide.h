// pt_wide.h -- construction of the traversal layout: per-node routines shared by the GPU builder
// (pt_build.hip) and a host-side breadth-first collapse (plain C++) that CPU tests check it against.
#pragma once
#include "../../include/pt_hip.h"
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <limits>
#include <vector>

// ---------------------------------------------------------------------------------------------
// Traversal layout: 4-wide nodes with child boxes quantised to 8 bits on a per-node grid (64 bytes
// for four children instead of 64 bytes for two).  The traversal kernels are bound by the number of
// 16-byte-per-lane loads the texture path has to serve for scattered node fetches, so bytes per
// child are what counts.  A wide node stands for a node of a binary tree (the binned-SAH tree of
// pt_sah.h for a BLAS, the Morton tree for a TLAS); its entries
// are found by opening that node and then, while fewer than four entries exist, the inner entry
// with the largest surface area.  Quantisation only ever grows a box
// (checked with the same fma the kernels use to expand it), so the set of triangles a ray is tested
// against can grow but the closest hit cannot change.
struct PtWide4 {
    float origin[3];
    float scale_x;          // grid steps are powers of two (kept as floats: no unpacking in the kernels)
    int32_t child[4];       // >= 0 wide node, < 0 leaf (~c = first << 3 | count - 1), INT32_MIN empty
    uint32_t q[6];          // lo.x, lo.y, lo.z, hi.x, hi.y, hi.z; byte k of each word belongs to child k
    float scale_y, scale_z;
};
static_assert(sizeof(PtWide4) == 64, "wide node must be 64 bytes");

// The same node with its planes as floats, one plane of the four children per 16 bytes: a kernel that picks the
// near and far plane vectors by address (the ray's sign offsets the load) needs neither byte conversions nor selects.
struct PtWideF {
    float lo_x[4], hi_x[4], lo_y[4], hi_y[4], lo_z[4], hi_z[4];    // an empty child is (+inf, -inf): no ray interval overlaps it
    int32_t child[4];
    uint32_t pad[4];
};
static_assert(sizeof(PtWideF) == 128, "float wide node must be 128 bytes");

#if defined(__HIPCC__)
#define PT_HD __host__ __device__
#else
#define PT_HD
#endif
#define PT_WIDE_EMPTY ((int32_t)0x80000000)
// Slot e of a pool node is unused: its box is inverted (a child's low byte never exceeds its high byte).
PT_HD static inline bool pt_wide_slot_unused(const PtWide4& w, int e) { return ((w.q[0] >> (8 * e)) & 255u) > ((w.q[3] >> (8 * e)) & 255u); }

struct PtWideEntry { int32_t ref; float lo[3], hi[3]; };

PT_HD static inline float pt_grid_scale(uint32_t exp_byte)
{
    union { uint32_t u; float f; } c;
    c.u = exp_byte << 23;
    return c.f;
}
PT_HD static inline float pt_grid_plane(uint32_t q, float scale, float origin)
{
    return __builtin_fmaf((float)q, scale, origin);
}

// Entries of the wide node that stands for binary node `node`: its children, then, while fewer
// than four entries exist, the inner entry with the largest surface area is replaced by its own
// children.  Returns the number of entries (1..4).
PT_HD static inline int pt_wide_open(const PtBvhNode* n2, uint32_t node, PtWideEntry ent[4])
{
    int ne = 0;
#define PT_WIDE_ADD(c, lo_, hi_) { ent[ne].ref = (c); for (int k_ = 0; k_ < 3; ++k_) { ent[ne].lo[k_] = (lo_)[k_]; ent[ne].hi[k_] = (hi_)[k_]; } ++ne; }
    const PtBvhNode& nd = n2[node];
    if (nd.c0 != PT_WIDE_EMPTY) PT_WIDE_ADD(nd.c0, nd.lo0, nd.hi0)
    if (nd.c1 != PT_WIDE_EMPTY) PT_WIDE_ADD(nd.c1, nd.lo1, nd.hi1)
    while (ne < 4) {
        int pick = -1; float pick_area = -1.0f;
        for (int e = 0; e < ne; ++e) {
            if (ent[e].ref < 0) continue;
            const float dx = ent[e].hi[0] - ent[e].lo[0], dy = ent[e].hi[1] - ent[e].lo[1], dz = ent[e].hi[2] - ent[e].lo[2];
            const float area = dx * dy + dy * dz + dz * dx;
            if (area > pick_area) { pick_area = area; pick = e; }
        }
        if (pick < 0) break;
        const PtBvhNode& ch = n2[ent[pick].ref];
        ent[pick] = ent[--ne];
        if (ch.c0 != PT_WIDE_EMPTY) PT_WIDE_ADD(ch.c0, ch.lo0, ch.hi0)
        if (ch.c1 != PT_WIDE_EMPTY) PT_WIDE_ADD(ch.c1, ch.lo1, ch.hi1)
    }
#undef PT_WIDE_ADD
    return ne;
}

// Fills origin, grid steps and quantised planes of `w` (zeroed by the caller) from the entries;
// child references are left to the caller.  Every dequantised box contains its entry's box.
PT_HD static inline void pt_wide_quantise(const PtWideEntry* ent, int ne, PtWide4& w)
{
    float lo[3], hi[3];
    for (int k = 0; k < 3; ++k) { lo[k] = ent[0].lo[k]; hi[k] = ent[0].hi[k]; }
    for (int e = 1; e < ne; ++e) for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], ent[e].lo[k]); hi[k] = fmaxf(hi[k], ent[e].hi[k]); }
    float scales[3];
    for (int k = 0; k < 3; ++k) {
        w.origin[k] = lo[k];
        // smallest power-of-two step with 255 steps covering the extent; grown until every plane fits
        int e2; const float ext = hi[k] - lo[k];
        if (!(ext > 0.0f)) e2 = -126; else { (void)frexpf(ext / 255.0f, &e2); }
        int eb = e2 + 127; if (eb < 1) eb = 1; if (eb > 254) eb = 254;
        for (;; ++eb) {
            const float scale = pt_grid_scale((uint32_t)eb);
            bool ok = true;
            for (int e = 0; e < ne && ok; ++e) {
                const double qh = ceil(((double)ent[e].hi[k] - (double)lo[k]) / (double)scale);
                if (qh > 255.0) ok = false;
                else { uint32_t q = (uint32_t)(qh < 0 ? 0 : qh); while (q <= 255 && pt_grid_plane(q, scale, lo[k]) < ent[e].hi[k]) ++q; if (q > 255) ok = false; }
            }
            if (ok || eb >= 254) break;
        }
        scales[k] = pt_grid_scale((uint32_t)eb);
    }
    w.scale_x = scales[0]; w.scale_y = scales[1]; w.scale_z = scales[2];
    for (int e = 0; e < ne; ++e) {
        for (int k = 0; k < 3; ++k) {
            const float scale = scales[k];
            const double ql = floor(((double)ent[e].lo[k] - (double)lo[k]) / (double)scale);
            uint32_t q_lo = (uint32_t)(ql < 0 ? 0 : (ql > 255 ? 255 : ql));
            while (q_lo > 0 && pt_grid_plane(q_lo, scale, lo[k]) > ent[e].lo[k]) --q_lo;
            const double qh = ceil(((double)ent[e].hi[k] - (double)lo[k]) / (double)scale);
            uint32_t q_hi = (uint32_t)(qh < 0 ? 0 : (qh > 255 ? 255 : qh));
            while (q_hi < 255 && pt_grid_plane(q_hi, scale, lo[k]) < ent[e].hi[k]) ++q_hi;
            w.q[k] |= q_lo << (8 * e);
            w.q[3 + k] |= q_hi << (8 * e);
        }
    }
}

// Leaf reference with its triangle position moved by tri_base.
PT_HD static inline int32_t pt_wide_leaf_rebased(int32_t ref, uint32_t tri_base)
{
    const uint32_t code = ~(uint32_t)ref;
    return (int32_t)~((((code >> 3) + tri_base) << 3) | (code & 7u));
}

// Host collapse of one binary BVH into wide nodes appended to `out`, breadth first: the wide node
// of a binary node is numbered when its parent wide node is written, level by level, which is the
// numbering the GPU builder reproduces with a scan per level.  Inner references are rebased by the
// position of the first appended node, leaf references by tri_base.  Returns the depth of the wide
// tree (a single node has depth 1).
static inline uint32_t pt_collapse_to_wide(const std::vector<PtBvhNode>& n2, uint32_t tri_base, std::vector<PtWide4>& out)
{
    const uint32_t base = (uint32_t)out.size();
    std::vector<uint32_t> todo { 0 };            // binary nodes that become wide nodes, in breadth-first order
    std::vector<uint32_t> depth_of { 1 };
    uint32_t max_depth = 1;
    for (size_t head = 0; head < todo.size(); ++head) {
        PtWideEntry ent[4];
        const int ne = pt_wide_open(n2.data(), todo[head], ent);
        PtWide4 w; memset(&w, 0, sizeof w);
        pt_wide_quantise(ent, ne, w);
        for (int e = 0; e < 4; ++e) w.child[e] = PT_WIDE_EMPTY;
        for (int e = 0; e < ne; ++e) {
            if (ent[e].ref >= 0) {
                w.child[e] = (int32_t)(base + (uint32_t)todo.size());
                todo.push_back((uint32_t)ent[e].ref);
                depth_of.push_back(depth_of[head] + 1);
                max_depth = depth_of[head] + 1 > max_depth ? depth_of[head] + 1 : max_depth;
            } else {
                w.child[e] = pt_wide_leaf_rebased(ent[e].ref, tri_base);
            }
        }
        out.push_back(w);
    }
    return max_depth;
}
