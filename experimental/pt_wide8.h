// pt_wide8.h -- the traversal layout: 8-wide nodes with implicit child references, and the per-node
// routines shared by the GPU builder (pt_build.hip) and a host breadth-first collapse that CPU tests
// check it against.
//
// What the traversal kernels pay for is vector instructions per visited child and bytes per visited
// node, so a node keeps both small:
//   * eight children in 80 bytes (five 16-byte loads): child boxes quantised to 8 bits on the node's
//     own power-of-two grid, a plane is fma(byte, 2^e, origin);
//   * no stored child references: the inner children of a node are consecutive nodes starting at
//     child_base, its leaf children consecutive leaf items (one triangle of a BLAS, one instance of a
//     TLAS per leaf child) starting at leaf_base, both in slot order, so a child is base + the number
//     of set mask bits below its slot;
//   * no sorting at run time: the builder puts each child into the slot whose three bits say on which
//     side of the node it lies (bit a set: towards +axis a), and a ray visits the slots in the order
//     slot XOR (7 - octant of its direction), far to near from bit 0 up: the hit children of a node are
//     one 8-bit mask in that order, kept as one stack entry and taken bit by bit.
// Quantisation only ever grows a box (checked with the same fma the kernels use), and every leaf
// child's box contains its primitive's padded box, so the set of primitives a ray is tested against
// can grow but the closest hit cannot change.
#pragma once
#include "pt_wide.h"

struct PtWide8 {
    float origin[3];
    uint32_t exp_imask;      // grid exponents ex | ey << 8 | ez << 16 (step = 2^(e - 127)), inner-child mask << 24
    uint32_t child_base;     // node index of the first inner child
    uint32_t leaf_base;      // position of the first leaf child's item
    uint32_t lmask;          // leaf-child mask (low 8 bits)
    uint32_t pad;
    uint8_t qlo[3][8];       // [axis][slot]
    uint8_t qhi[3][8];
};
static_assert(sizeof(PtWide8) == 80, "wide node must be 80 bytes");

// Priority of slot s for a ray whose direction has octant oct (bit a set: d[a] < 0): the slot with
// the highest priority is visited first.
PT_HD static inline uint32_t pt_w8_priority(uint32_t slot, uint32_t oct) { return slot ^ (7u ^ oct); }

// Entries of the wide node that stands for binary node `node`: its children, then, while fewer than
// eight entries exist, the inner entry with the largest surface area is replaced by its own
// children.  Leaf references of the binary tree must hold one primitive each.  Returns the number
// of entries (1..8), or -1 if a leaf reference holds several primitives.
PT_HD static inline int pt_w8_open(const PtBvhNode* n2, uint32_t node, PtWideEntry ent[8])
{
    int ne = 0;
#define PT_W8_ADD(c, lo_, hi_) { ent[ne].ref = (c); for (int k_ = 0; k_ < 3; ++k_) { ent[ne].lo[k_] = (lo_)[k_]; ent[ne].hi[k_] = (hi_)[k_]; } ++ne; }
    const PtBvhNode& nd = n2[node];
    if (nd.c0 != PT_WIDE_EMPTY) PT_W8_ADD(nd.c0, nd.lo0, nd.hi0)
    if (nd.c1 != PT_WIDE_EMPTY) PT_W8_ADD(nd.c1, nd.lo1, nd.hi1)
    while (ne < 8) {
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
        if (ch.c0 != PT_WIDE_EMPTY) PT_W8_ADD(ch.c0, ch.lo0, ch.hi0)
        if (ch.c1 != PT_WIDE_EMPTY) PT_W8_ADD(ch.c1, ch.lo1, ch.hi1)
    }
#undef PT_W8_ADD
    for (int e = 0; e < ne; ++e) if (ent[e].ref < 0 && ((~(uint32_t)ent[e].ref) & 7u) != 0u) return -1;
    return ne;
}

// Slot of every entry: entry c likes slot s by the sum over the axes of +-(centre of c - centre of
// the node), + where bit a of s is set; pairs are fixed greedily, best first (ties: lower entry,
// then lower slot).  slot_of[c] in 0..7, all different.
PT_HD static inline void pt_w8_assign_slots(const PtWideEntry* ent, int ne, int slot_of[8])
{
    float lo[3], hi[3];
    for (int k = 0; k < 3; ++k) { lo[k] = ent[0].lo[k]; hi[k] = ent[0].hi[k]; }
    for (int e = 1; e < ne; ++e) for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], ent[e].lo[k]); hi[k] = fmaxf(hi[k], ent[e].hi[k]); }
    float rel[8][3];
    for (int e = 0; e < ne; ++e) for (int k = 0; k < 3; ++k) rel[e][k] = (ent[e].lo[k] + ent[e].hi[k]) - (lo[k] + hi[k]);
    bool entry_done[8], slot_done[8];
    for (int k = 0; k < 8; ++k) { entry_done[k] = false; slot_done[k] = false; slot_of[k] = -1; }
    for (int round = 0; round < ne; ++round) {
        int bc = -1, bs = -1; float best = 0.0f;
        for (int c = 0; c < ne; ++c) {
            if (entry_done[c]) continue;
            for (int s = 0; s < 8; ++s) {
                if (slot_done[s]) continue;
                const float cost = ((s & 1) ? rel[c][0] : -rel[c][0]) + ((s & 2) ? rel[c][1] : -rel[c][1]) + ((s & 4) ? rel[c][2] : -rel[c][2]);
                if (bc < 0 || cost > best) { best = cost; bc = c; bs = s; }
            }
        }
        entry_done[bc] = true; slot_done[bs] = true; slot_of[bc] = bs;
    }
}

// Fills origin, exponents and quantised planes of `w` (zeroed by the caller) from the entries in their
// slots; masks and bases are left to the caller.  Every dequantised box contains its entry's box.
// Empty slots keep the inverted box lo = 255, hi = 0.
PT_HD static inline void pt_w8_quantise(const PtWideEntry* ent, int ne, const int* slot_of, PtWide8& w)
{
    float lo[3], hi[3];
    for (int k = 0; k < 3; ++k) { lo[k] = ent[0].lo[k]; hi[k] = ent[0].hi[k]; }
    for (int e = 1; e < ne; ++e) for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], ent[e].lo[k]); hi[k] = fmaxf(hi[k], ent[e].hi[k]); }
    uint32_t ebs[3];
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
        ebs[k] = (uint32_t)eb;
    }
    w.exp_imask = ebs[0] | (ebs[1] << 8) | (ebs[2] << 16);
    for (int k = 0; k < 3; ++k) for (int s = 0; s < 8; ++s) { w.qlo[k][s] = 255; w.qhi[k][s] = 0; }
    for (int e = 0; e < ne; ++e) {
        for (int k = 0; k < 3; ++k) {
            const float scale = pt_grid_scale(ebs[k]);
            const double ql = floor(((double)ent[e].lo[k] - (double)lo[k]) / (double)scale);
            uint32_t q_lo = (uint32_t)(ql < 0 ? 0 : (ql > 255 ? 255 : ql));
            while (q_lo > 0 && pt_grid_plane(q_lo, scale, lo[k]) > ent[e].lo[k]) --q_lo;
            const double qh = ceil(((double)ent[e].hi[k] - (double)lo[k]) / (double)scale);
            uint32_t q_hi = (uint32_t)(qh < 0 ? 0 : (qh > 255 ? 255 : qh));
            while (q_hi < 255 && pt_grid_plane(q_hi, scale, lo[k]) < ent[e].hi[k]) ++q_hi;
            w.qlo[k][slot_of[e]] = (uint8_t)q_lo;
            w.qhi[k][slot_of[e]] = (uint8_t)q_hi;
        }
    }
}

// The whole node of binary node `node` except its two bases: entries opened, put into slots,
// quantised, masks set.  binary_ref[s] = the binary tree's reference of the child in slot s (inner:
// node index; leaf: ~(position << 3)), PT_WIDE_EMPTY for an empty slot.  Returns the number of
// entries, or -1 (see pt_w8_open).
PT_HD static inline int pt_w8_make(const PtBvhNode* n2, uint32_t node, PtWide8& w, int32_t binary_ref[8])
{
    PtWideEntry ent[8];
    const int ne = pt_w8_open(n2, node, ent);
    if (ne < 0) return -1;
    int slot_of[8];
    pt_w8_assign_slots(ent, ne, slot_of);
    w.child_base = w.leaf_base = w.lmask = w.pad = 0u;
    pt_w8_quantise(ent, ne, slot_of, w);
    uint32_t imask = 0u, lmask = 0u;
    for (int s = 0; s < 8; ++s) binary_ref[s] = PT_WIDE_EMPTY;
    for (int e = 0; e < ne; ++e) {
        binary_ref[slot_of[e]] = ent[e].ref;
        if (ent[e].ref >= 0) imask |= 1u << slot_of[e]; else lmask |= 1u << slot_of[e];
    }
    w.exp_imask |= imask << 24;
    w.lmask = lmask;
    return ne;
}

PT_HD static inline uint32_t pt_w8_imask(const PtWide8& w) { return w.exp_imask >> 24; }
PT_HD static inline uint32_t pt_popc8(uint32_t m) { m = (m & 0x55u) + ((m >> 1) & 0x55u); m = (m & 0x33u) + ((m >> 2) & 0x33u); return (m & 0x0Fu) + (m >> 4); }

// Host collapse of one binary BVH into wide nodes appended to `out`, breadth first: the inner children
// of a node are numbered when the node is written, level by level and in slot order, which is the
// numbering the GPU builder reproduces with a scan per level; leaf items are numbered the same way.
// child_base is rebased by the position of the first appended node, leaf_base by leaf_first;
// order[j] = position (in the binary tree's leaf order) of leaf item j.  Returns the depth of the wide
// tree (a single node has depth 1), 0 if a leaf reference of the binary tree holds several primitives.
static inline uint32_t pt_collapse_to_wide8(const std::vector<PtBvhNode>& n2, uint32_t leaf_first, std::vector<PtWide8>& out, std::vector<uint32_t>& order)
{
    const uint32_t base = (uint32_t)out.size();
    std::vector<uint32_t> todo { 0 };
    std::vector<uint32_t> depth_of { 1 };
    uint32_t max_depth = 1;
    for (size_t head = 0; head < todo.size(); ++head) {
        PtWide8 w; memset(&w, 0, sizeof w);
        int32_t ref[8];
        if (pt_w8_make(n2.data(), todo[head], w, ref) < 0) return 0;
        w.child_base = base + (uint32_t)todo.size();
        w.leaf_base = leaf_first + (uint32_t)order.size();
        for (int s = 0; s < 8; ++s) {
            if (ref[s] == PT_WIDE_EMPTY) continue;
            if (ref[s] >= 0) {
                todo.push_back((uint32_t)ref[s]);
                depth_of.push_back(depth_of[head] + 1);
                max_depth = depth_of[head] + 1 > max_depth ? depth_of[head] + 1 : max_depth;
            } else order.push_back((~(uint32_t)ref[s]) >> 3);
        }
        out.push_back(w);
    }
    return max_depth;
}
