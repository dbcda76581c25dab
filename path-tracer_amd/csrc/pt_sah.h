// pt_sah.h -- top-down binned-SAH tree for the traversal layout: the per-node decisions shared by the
// GPU builder (pt_build.hip, one tree level per pass) and a host breadth-first twin that CPU-side
// tests compare it with.  The tree is only ever used through its boxes (every box is the exact union
// of the padded primitive boxes below it), so it changes which boxes a ray tests, never what it hits;
// the canonical Morton tree stays the structure that is compared with the oracle.
//
// A node over the primitive range [a, b) bins the primitives by centroid into PT_SAH_BINS bins per
// axis (the bins span the range's box: binning over the centroid bounds instead costs a pass of
// heavily contended atomics per level and gave the same tree cost), takes the split with the
// smallest area(L) * n(L) + area(R) * n(R) (first minimum in axis-major, bin-ascending order) and
// partitions the range stably.  A range of at most max_leaf
// primitives becomes a leaf when n * area <= area + best (the root never does); a range whose
// centroids coincide is split by position.
#pragma once
#include "pt_wide.h"
#include <algorithm>

#define PT_SAH_BINS 16

struct PtSahTask {                 // one range of the current level
    uint32_t a, b;                 // primitive positions
    int32_t parent;                // node that holds the reference to this range (-1: root)
    uint32_t side;                 // 0: parent's c0, 1: c1
    float lo[3], hi[3];            // box of the range
    // decision
    int32_t kind;                  // 0 leaf, 1 binned split, 2 split by position
    int32_t axis, bin;
    float clo, scale;              // bin = (centroid[axis] - clo) * scale
    uint32_t node;                 // index of the inner node written for this range
    float lo0[3], hi0[3], lo1[3], hi1[3];
};

struct PtSahBinsF {                // bins of one range, decoded
    float lo[3][PT_SAH_BINS][3], hi[3][PT_SAH_BINS][3];
    uint32_t cnt[3][PT_SAH_BINS];
};

PT_HD static inline float pt_sah_area(const float lo[3], const float hi[3])
{
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return (dx * dy + dy * dz) + dz * dx;
}
PT_HD static inline float pt_sah_centroid(const float* box, int axis) { return 0.5f * (box[axis] + box[3 + axis]); }
PT_HD static inline float pt_sah_scale(float clo, float chi) { return (float)PT_SAH_BINS / (chi - clo); }
PT_HD static inline int pt_sah_bin_of(const float* box, int axis, float clo, float scale)
{
    int q = (int)((pt_sah_centroid(box, axis) - clo) * scale);
    return q < 0 ? 0 : (q >= PT_SAH_BINS ? PT_SAH_BINS - 1 : q);
}

// Second half of a range's decision, given the cheapest split (best_ax < 0: none exists; cost `best`,
// children's boxes lo0..hi1): leaf, binned split or split by position.  Shared by the serial search
// below and the GPU builder's wave-parallel search, so both decide alike.
PT_HD static inline void pt_sah_finish(PtSahTask& t, int best_ax, int best_bin, float best, const float lo0[3], const float hi0[3],
                                       const float lo1[3], const float hi1[3], const float cb[6], uint32_t max_leaf)
{
    const uint32_t n = t.b - t.a;
    const bool have = best_ax >= 0;
    const bool can_leaf = t.parent >= 0 && n <= max_leaf;
    const float A = pt_sah_area(t.lo, t.hi);
    if (n == 1 || (can_leaf && (!have || (float)n * A <= A + best))) { t.kind = 0; return; }
    if (!have) { t.kind = 2; return; }            // child boxes of a positional split are filled by the caller
    t.kind = 1; t.axis = best_ax; t.bin = best_bin;
    t.clo = cb[best_ax]; t.scale = pt_sah_scale(cb[best_ax], cb[3 + best_ax]);
    for (int k = 0; k < 3; ++k) { t.lo0[k] = lo0[k]; t.hi0[k] = hi0[k]; t.lo1[k] = lo1[k]; t.hi1[k] = hi1[k]; }
}

// Fills the decision fields of `t` from its bins; cb (lo[3], hi[3]) is the interval the bins span.
PT_HD static inline void pt_sah_decide(PtSahTask& t, const PtSahBinsF& B, const float cb[6], uint32_t max_leaf)
{
    const uint32_t n = t.b - t.a;
    const float inf = __builtin_inff();
    float best = inf; int best_ax = -1, best_bin = 0;
    if (n > 1) {
        for (int ax = 0; ax < 3; ++ax) {
            if (!(cb[3 + ax] > cb[ax])) continue;
            // suffix unions
            float rl[PT_SAH_BINS][3], rh[PT_SAH_BINS][3]; uint32_t rn[PT_SAH_BINS];
            float tl[3] = { inf, inf, inf }, th[3] = { -inf, -inf, -inf }; uint32_t tn = 0;
            for (int q = PT_SAH_BINS - 1; q >= 1; --q) {
                tn += B.cnt[ax][q];
                for (int k = 0; k < 3; ++k) { tl[k] = fminf(tl[k], B.lo[ax][q][k]); th[k] = fmaxf(th[k], B.hi[ax][q][k]); rl[q][k] = tl[k]; rh[q][k] = th[k]; }
                rn[q] = tn;
            }
            for (int k = 0; k < 3; ++k) { tl[k] = inf; th[k] = -inf; }
            tn = 0;
            for (int q = 0; q < PT_SAH_BINS - 1; ++q) {
                tn += B.cnt[ax][q];
                for (int k = 0; k < 3; ++k) { tl[k] = fminf(tl[k], B.lo[ax][q][k]); th[k] = fmaxf(th[k], B.hi[ax][q][k]); }
                if (tn == 0 || rn[q + 1] == 0) continue;
                const float c = pt_sah_area(tl, th) * (float)tn + pt_sah_area(rl[q + 1], rh[q + 1]) * (float)rn[q + 1];
                if (c < best) { best = c; best_ax = ax; best_bin = q; }
            }
        }
    }
    float lo0[3], hi0[3], lo1[3], hi1[3];
    for (int k = 0; k < 3; ++k) { lo0[k] = inf; hi0[k] = -inf; lo1[k] = inf; hi1[k] = -inf; }
    if (best_ax >= 0)
        for (int q = 0; q < PT_SAH_BINS; ++q) {
            if (B.cnt[best_ax][q] == 0) continue;
            float* lo = q <= best_bin ? lo0 : lo1; float* hi = q <= best_bin ? hi0 : hi1;
            for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], B.lo[best_ax][q][k]); hi[k] = fmaxf(hi[k], B.hi[best_ax][q][k]); }
        }
    pt_sah_finish(t, best_ax, best_bin, best, lo0, hi0, lo1, hi1, cb, max_leaf);
}

// true: the primitive at position i goes to the left child of the split range t
PT_HD static inline bool pt_sah_goes_left(const PtSahTask& t, const float* box, uint32_t i)
{
    if (t.kind == 1) return pt_sah_bin_of(box, t.axis, t.clo, t.scale) <= t.bin;
    return i - t.a < (t.b - t.a) / 2;
}

// Child boxes of a positional split (ids: primitive at each position).
PT_HD static inline void pt_sah_positional_boxes(PtSahTask& t, const float* boxes, const uint32_t* ids)
{
    const float inf = __builtin_inff();
    for (int k = 0; k < 3; ++k) { t.lo0[k] = inf; t.hi0[k] = -inf; t.lo1[k] = inf; t.hi1[k] = -inf; }
    const uint32_t mid = t.a + (t.b - t.a) / 2;
    for (uint32_t i = t.a; i < t.b; ++i) {
        const float* x = boxes + 6 * (size_t)ids[i];
        float* lo = i < mid ? t.lo0 : t.lo1; float* hi = i < mid ? t.hi0 : t.hi1;
        for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], x[k]); hi[k] = fmaxf(hi[k], x[3 + k]); }
    }
}

PT_HD static inline int32_t pt_sah_leaf_ref(uint32_t a, uint32_t n) { return ~(int32_t)((a << 3) | (n - 1u)); }

// Host twin: same decisions, same level order, same numbering (inner ranges of a level in range order).
// boxes: n x 6 padded primitive boxes; ids_out: primitive at each leaf position; nodes_out[0] is the root.
static inline void pt_sah_build_host(const float* boxes, uint32_t n, uint32_t max_leaf, const float root_lo[3], const float root_hi[3],
                                     std::vector<PtBvhNode>& nodes_out, std::vector<uint32_t>& ids_out)
{
    nodes_out.clear(); ids_out.resize(n);
    for (uint32_t i = 0; i < n; ++i) ids_out[i] = i;
    std::vector<PtSahTask> level(1), next;
    memset(&level[0], 0, sizeof(PtSahTask));
    level[0].a = 0; level[0].b = n; level[0].parent = -1;
    for (int k = 0; k < 3; ++k) { level[0].lo[k] = root_lo[k]; level[0].hi[k] = root_hi[k]; }
    const float inf = __builtin_inff();
    while (!level.empty()) {
        next.clear();
        for (PtSahTask& t : level) {
            PtSahBinsF B; float cb[6];
            for (int k = 0; k < 3; ++k) { cb[k] = t.lo[k]; cb[3 + k] = t.hi[k]; }     // bins span the range's box
            for (int ax = 0; ax < 3; ++ax) for (int q = 0; q < PT_SAH_BINS; ++q) { B.cnt[ax][q] = 0; for (int k = 0; k < 3; ++k) { B.lo[ax][q][k] = inf; B.hi[ax][q][k] = -inf; } }
            for (int ax = 0; ax < 3; ++ax) {
                if (!(cb[3 + ax] > cb[ax])) continue;
                const float sc = pt_sah_scale(cb[ax], cb[3 + ax]);
                for (uint32_t i = t.a; i < t.b; ++i) {
                    const float* x = boxes + 6 * (size_t)ids_out[i];
                    const int q = pt_sah_bin_of(x, ax, cb[ax], sc);
                    B.cnt[ax][q]++;
                    for (int k = 0; k < 3; ++k) { B.lo[ax][q][k] = fminf(B.lo[ax][q][k], x[k]); B.hi[ax][q][k] = fmaxf(B.hi[ax][q][k], x[3 + k]); }
                }
            }
            pt_sah_decide(t, B, cb, max_leaf);
            if (t.kind == 2) pt_sah_positional_boxes(t, boxes, ids_out.data());
        }
        // number the inner ranges of this level in order, write their nodes and their parents' references
        for (PtSahTask& t : level) {
            int32_t ref;
            if (t.kind == 0) ref = pt_sah_leaf_ref(t.a, t.b - t.a);
            else {
                t.node = (uint32_t)nodes_out.size();
                PtBvhNode nd; memset(&nd, 0, sizeof nd);
                for (int k = 0; k < 3; ++k) { nd.lo0[k] = t.lo0[k]; nd.hi0[k] = t.hi0[k]; nd.lo1[k] = t.lo1[k]; nd.hi1[k] = t.hi1[k]; }
                nodes_out.push_back(nd);
                ref = (int32_t)t.node;
            }
            if (t.parent >= 0) { if (t.side) nodes_out[t.parent].c1 = ref; else nodes_out[t.parent].c0 = ref; }
        }
        for (PtSahTask& t : level) {
            if (t.kind == 0) continue;
            std::vector<uint32_t> left, right;
            for (uint32_t i = t.a; i < t.b; ++i) {
                const uint32_t id = ids_out[i];
                (pt_sah_goes_left(t, boxes + 6 * (size_t)id, i) ? left : right).push_back(id);
            }
            const uint32_t mid = t.a + (uint32_t)left.size();
            std::copy(left.begin(), left.end(), ids_out.begin() + t.a);
            std::copy(right.begin(), right.end(), ids_out.begin() + mid);
            PtSahTask c0, c1; memset(&c0, 0, sizeof c0); memset(&c1, 0, sizeof c1);
            c0.a = t.a; c0.b = mid; c0.parent = (int32_t)t.node; c0.side = 0;
            c1.a = mid; c1.b = t.b; c1.parent = (int32_t)t.node; c1.side = 1;
            for (int k = 0; k < 3; ++k) { c0.lo[k] = t.lo0[k]; c0.hi[k] = t.hi0[k]; c1.lo[k] = t.lo1[k]; c1.hi[k] = t.hi1[k]; }
            next.push_back(c0); next.push_back(c1);
        }
        level.swap(next);
    }
}

// Full-sweep SAH tree over a few boxes (the instances of a TLAS): every object partition along the three axes by
// centroid order is priced with area(L) * n(L) + area(R) * n(R) and the cheapest taken (first minimum in axis-major,
// position-ascending order; centroids are ordered by (value, index)); one box per leaf.  The Morton tree that is
// compared with the CPU port orders a handful of large, overlapping instance boxes (the walls of a room, the parts of
// one object) more or less arbitrarily; this tree is what the traversal layout of a TLAS is derived from, and like
// every tree over the same boxes it changes which boxes a ray tests, never what it hits.
// boxes: n x 6; nodes_out[0] is the root (n > 1); order_out[p] = box at leaf position p.
static inline void pt_sweep_sah_build_host(const float* boxes, uint32_t n, std::vector<PtBvhNode>& nodes_out, std::vector<uint32_t>& order_out)
{
    nodes_out.clear(); order_out.resize(n);
    for (uint32_t i = 0; i < n; ++i) order_out[i] = i;
    if (n < 2) return;
    const float inf = __builtin_inff();
    struct Range { uint32_t a, b; int32_t parent; uint32_t side; };
    std::vector<Range> todo { Range { 0, n, -1, 0 } };
    std::vector<uint32_t> sorted, best_sorted;
    std::vector<float> suffix;
    for (size_t head = 0; head < todo.size(); ++head) {
        const Range r = todo[head];
        const uint32_t m = r.b - r.a;
        int32_t ref;
        if (m == 1) ref = pt_sah_leaf_ref(r.a, 1);
        else {
            float best = inf; uint32_t best_k = m / 2; bool have = false;
            for (int ax = 0; ax < 3; ++ax) {
                sorted.assign(order_out.begin() + r.a, order_out.begin() + r.b);
                std::stable_sort(sorted.begin(), sorted.end(), [&](uint32_t x, uint32_t y) {
                    const float cx = pt_sah_centroid(boxes + 6 * (size_t)x, ax), cy = pt_sah_centroid(boxes + 6 * (size_t)y, ax);
                    return cx < cy || (cx == cy && x < y);
                });
                suffix.assign(m + 1, 0.0f);
                float lo[3] = { inf, inf, inf }, hi[3] = { -inf, -inf, -inf };
                for (uint32_t k = m; k-- > 1;) {
                    const float* x = boxes + 6 * (size_t)sorted[k];
                    for (int c = 0; c < 3; ++c) { lo[c] = fminf(lo[c], x[c]); hi[c] = fmaxf(hi[c], x[3 + c]); }
                    suffix[k] = pt_sah_area(lo, hi);
                }
                for (int c = 0; c < 3; ++c) { lo[c] = inf; hi[c] = -inf; }
                for (uint32_t k = 1; k < m; ++k) {          // the first k boxes go left
                    const float* x = boxes + 6 * (size_t)sorted[k - 1];
                    for (int c = 0; c < 3; ++c) { lo[c] = fminf(lo[c], x[c]); hi[c] = fmaxf(hi[c], x[3 + c]); }
                    const float cost = pt_sah_area(lo, hi) * (float)k + suffix[k] * (float)(m - k);
                    if (cost < best) { best = cost; best_k = k; best_sorted = sorted; have = true; }
                }
            }
            if (have) std::copy(best_sorted.begin(), best_sorted.end(), order_out.begin() + r.a);
            const uint32_t mid = r.a + best_k;
            PtBvhNode nd; memset(&nd, 0, sizeof nd);
            for (int c = 0; c < 3; ++c) { nd.lo0[c] = inf; nd.hi0[c] = -inf; nd.lo1[c] = inf; nd.hi1[c] = -inf; }
            for (uint32_t i = r.a; i < r.b; ++i) {
                const float* x = boxes + 6 * (size_t)order_out[i];
                float* lo = i < mid ? nd.lo0 : nd.lo1; float* hi = i < mid ? nd.hi0 : nd.hi1;
                for (int c = 0; c < 3; ++c) { lo[c] = fminf(lo[c], x[c]); hi[c] = fmaxf(hi[c], x[3 + c]); }
            }
            ref = (int32_t)nodes_out.size();
            nodes_out.push_back(nd);
            todo.push_back(Range { r.a, mid, ref, 0 });
            todo.push_back(Range { mid, r.b, ref, 1 });
        }
        if (r.parent >= 0) { if (r.side) nodes_out[r.parent].c1 = ref; else nodes_out[r.parent].c0 = ref; }
    }
}

// Expected box tests of a ray that enters the root, in the surface-area model: the areas of all child boxes relative to
// the root's.  What decides between two trees over the same boxes.
static inline double pt_tree_cost_host(const PtBvhNode* nodes, size_t n)
{
    if (n == 0) return 0.0;
    float lo[3], hi[3];
    for (int c = 0; c < 3; ++c) { lo[c] = fminf(nodes[0].lo0[c], nodes[0].lo1[c]); hi[c] = fmaxf(nodes[0].hi0[c], nodes[0].hi1[c]); }
    const double root = pt_sah_area(lo, hi);
    double sum = 0.0;
    for (size_t i = 0; i < n; ++i) sum += (double)pt_sah_area(nodes[i].lo0, nodes[i].hi0) + (double)pt_sah_area(nodes[i].lo1, nodes[i].hi1);
    return root > 0.0 ? sum / root : 0.0;
}
