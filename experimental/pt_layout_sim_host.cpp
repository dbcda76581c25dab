// pt_layout_sim_host.cpp -- host-side collapse into the 8-wide layout (pt_wide8.h) and traversal simulators of the
// 8-wide and 4-wide layouts with pruning by the closest hit: what the layout comparison (sim_layouts.py) counts
// node visits and triangle tests with.  Not part of libpt_hip.so or of the default build.
#include "pt_wide.h"
#include "pt_wide8.h"

namespace {
const float kDirEps = 8.271806125530277e-25f;   // 2^-80
inline float safe_rcp(float d) { float s = fabsf(d) > kDirEps ? d : copysignf(kDirEps, d); return 1.0f / s; }
}

extern "C" {

// ---- 8-wide layout ----------------------------------------------------------------------------
// Collapses a binary BVH (one primitive per leaf reference) into 8-wide nodes; order_out[j] = leaf
// position of leaf item j.  Returns the number of wide nodes (0 if `cap` is too small or a leaf
// reference holds several primitives).
uint32_t pt_wide8_collapse_host(const PtBvhNode* nodes, uint32_t node_count, uint32_t leaf_first,
                                void* out, uint32_t cap, uint32_t* order_out, uint32_t* depth)
{
    std::vector<PtBvhNode> n2(nodes, nodes + node_count);
    std::vector<PtWide8> w; std::vector<uint32_t> order;
    const uint32_t d = pt_collapse_to_wide8(n2, leaf_first, w, order);
    if (depth) *depth = d;
    if (d == 0 || w.size() > cap) return 0;
    memcpy(out, w.data(), sizeof(PtWide8) * w.size());
    if (order_out) memcpy(order_out, order.data(), sizeof(uint32_t) * order.size());
    return (uint32_t)w.size();
}

namespace {
// The child boxes of one node a ray interval overlaps, as a mask over the slots.
inline uint32_t w8_hits(const PtWide8& w, const float inv[3], const float oo[3], float tlo, float thi)
{
    uint32_t m = 0;
    const uint32_t valid = pt_w8_imask(w) | w.lmask;
    for (int s = 0; s < 8; ++s) {
        if (!((valid >> s) & 1u)) continue;
        float tn = -std::numeric_limits<float>::infinity(), tf = std::numeric_limits<float>::infinity();
        for (int k = 0; k < 3; ++k) {
            const float sc = pt_grid_scale((w.exp_imask >> (8 * k)) & 255u);
            const float lo = pt_grid_plane(w.qlo[k][s], sc, w.origin[k]), hi = pt_grid_plane(w.qhi[k][s], sc, w.origin[k]);
            const float a = __builtin_fmaf(lo, inv[k], -oo[k]), b = __builtin_fmaf(hi, inv[k], -oo[k]);
            tn = fmaxf(tn, fminf(a, b)); tf = fminf(tf, fmaxf(a, b));
        }
        if (fmaxf(tn, tlo) <= fminf(tf, thi)) m |= 1u << s;
    }
    return m;
}
// Moeller-Trumbore in binary64: the simulations below only need a distance to prune with.
inline bool tri_t(const float* v, const float o[3], const float d[3], double& t)
{
    const double e1[3] = { (double)v[3] - v[0], (double)v[4] - v[1], (double)v[5] - v[2] }, e2[3] = { (double)v[6] - v[0], (double)v[7] - v[1], (double)v[8] - v[2] };
    const double p[3] = { d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0] };
    const double det = e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2];
    if (det == 0.0) return false;
    const double s[3] = { (double)o[0] - v[0], (double)o[1] - v[1], (double)o[2] - v[2] };
    const double u = (s[0] * p[0] + s[1] * p[1] + s[2] * p[2]) / det;
    if (u < 0.0 || u > 1.0) return false;
    const double q[3] = { s[1] * e1[2] - s[2] * e1[1], s[2] * e1[0] - s[0] * e1[2], s[0] * e1[1] - s[1] * e1[0] };
    const double vv = (d[0] * q[0] + d[1] * q[1] + d[2] * q[2]) / det;
    if (vv < 0.0 || u + vv > 1.0) return false;
    t = (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]) / det;
    return true;
}
}

// Leaf items of every leaf child whose chain of dequantised boxes the ray interval overlaps, without
// pruning by found hits.
uint32_t pt_wide8_candidates_host(const void* wide, uint32_t root, const float o[3], const float d[3],
                                  float tlo, float thi, uint32_t* out_items, uint32_t cap)
{
    const PtWide8* W = (const PtWide8*)wide;
    float inv[3], oo[3];
    for (int k = 0; k < 3; ++k) { inv[k] = safe_rcp(d[k]); oo[k] = o[k] * inv[k]; }
    std::vector<uint32_t> stack { root };
    uint32_t n = 0;
    while (!stack.empty()) {
        const PtWide8& w = W[stack.back()]; stack.pop_back();
        const uint32_t m = w8_hits(w, inv, oo, tlo, thi), im = pt_w8_imask(w);
        for (uint32_t s = 0; s < 8; ++s) {
            if (!((m >> s) & 1u)) continue;
            const uint32_t below = (1u << s) - 1u;
            if ((im >> s) & 1u) stack.push_back(w.child_base + pt_popc8(im & below));
            else { if (n < cap) out_items[n] = w.leaf_base + pt_popc8(w.lmask & below); ++n; }
        }
    }
    return n;
}

// Simulation of the kernels' traversal of one structure (no instancing) with pruning by the closest hit:
// slots in octant order, hit children kept as masks.  tris: 9 floats per leaf item.  stats[0] += nodes
// visited, [1] += triangles tested, [2] += largest stack depth seen (max), per call over all rays.
void pt_wide8_simulate_host(const void* wide, const float* tris, uint32_t n_rays, const float* o3, const float* d3,
                            float tmin, float tmax, uint64_t* stats, float* t_out)
{
    const PtWide8* W = (const PtWide8*)wide;
    struct G { uint32_t x, y; };
    for (uint32_t r = 0; r < n_rays; ++r) {
        const float* o = o3 + 3 * (size_t)r; const float* d = d3 + 3 * (size_t)r;
        float inv[3], oo[3];
        for (int k = 0; k < 3; ++k) { inv[k] = safe_rcp(d[k]); oo[k] = o[k] * inv[k]; }
        const uint32_t oct = (d[0] < 0 ? 1u : 0u) | (d[1] < 0 ? 2u : 0u) | (d[2] < 0 ? 4u : 0u), flip = 7u ^ oct;
        auto permute = [&](uint32_t m) { uint32_t p = 0; for (uint32_t s = 0; s < 8; ++s) if ((m >> s) & 1u) p |= 1u << (s ^ flip); return p; };
        double best = tmax;
        std::vector<G> stack;
        G cur { 0u, 0x80000000u };
        size_t deepest = 0;
        for (;;) {
            if (cur.y >= 0x01000000u) {
                const uint32_t p = 31u - (uint32_t)__builtin_clz(cur.y);
                cur.y &= ~(1u << p);
                const uint32_t slot = (p - 24u) ^ flip, im = cur.y & 0xFFu;
                const PtWide8& w = W[cur.x + pt_popc8(im & ((1u << slot) - 1u))];
                ++stats[0];
                const uint32_t m = w8_hits(w, inv, oo, tmin, (float)best), wi = pt_w8_imask(w);
                const uint32_t mi = permute(m & wi), ml = permute(m & w.lmask);
                const G gn { w.child_base, (mi << 24) | wi }, gl { w.leaf_base, (ml << 8) | w.lmask };
                const G rest = cur;
                G order[3]; int no = 0;
                if (ml) order[no++] = gl;
                if (mi) order[no++] = gn;
                if (rest.y >= 0x01000000u) order[no++] = rest;
                if (no == 0) { if (stack.empty()) break; cur = stack.back(); stack.pop_back(); continue; }
                cur = order[0];
                for (int k = no - 1; k >= 1; --k) stack.push_back(order[k]);
                deepest = stack.size() > deepest ? stack.size() : deepest;
            } else if (cur.y >= 0x100u) {
                const uint32_t p = 31u - (uint32_t)__builtin_clz(cur.y);
                cur.y &= ~(1u << p);
                const uint32_t slot = (p - 8u) ^ flip, lm = cur.y & 0xFFu;
                const uint32_t item = cur.x + pt_popc8(lm & ((1u << slot) - 1u));
                ++stats[1];
                double t;
                if (tri_t(tris + 9 * (size_t)item, o, d, t) && t > tmin && t < best) best = t;
            } else {
                if (stack.empty()) break;
                cur = stack.back(); stack.pop_back();
            }
        }
        if (deepest > stats[2]) stats[2] = deepest;
        if (t_out) t_out[r] = (float)best;
    }
}

// The same for the 4-wide layout (children ordered by entry distance, leaves of up to four triangles):
// tris hold 9 floats per sorted position.
// order: 0 children by entry distance (what the kernels do), 1 as stored, 2 along the axis on which the children's
// centres spread most, reversed for a ray that runs down that axis, 3 by the octant of the direction
// (sum over the axes of +-centre).
void pt_wide4_simulate_host(const void* wide, const float* tris, uint32_t n_rays, const float* o3, const float* d3,
                            float tmin, float tmax, uint64_t* stats, float* t_out, int order)
{
    const PtWide4* W = (const PtWide4*)wide;
    const int32_t EMPTY = std::numeric_limits<int32_t>::min();
    for (uint32_t r = 0; r < n_rays; ++r) {
        const float* o = o3 + 3 * (size_t)r; const float* d = d3 + 3 * (size_t)r;
        float inv[3], oo[3];
        for (int k = 0; k < 3; ++k) { inv[k] = safe_rcp(d[k]); oo[k] = o[k] * inv[k]; }
        double best = tmax;
        std::vector<int32_t> stack { 0 };
        size_t deepest = 0;
        while (!stack.empty()) {
            const int32_t ref = stack.back(); stack.pop_back();
            if (ref < 0) {
                const uint32_t code = ~(uint32_t)ref, first = code >> 3, cnt = (code & 7u) + 1u;
                for (uint32_t k = 0; k < cnt; ++k) {
                    ++stats[1];
                    double t;
                    if (tri_t(tris + 9 * (size_t)(first + k), o, d, t) && t > tmin && t < best) best = t;
                }
                continue;
            }
            ++stats[0];
            const PtWide4& w = W[ref];
            float tn[4]; int32_t rf[4]; int nh = 0;
            float ctr[4][3]; int nv = 0;
            for (int e = 0; e < 4; ++e) {
                if (w.child[e] == EMPTY) continue;
                for (int k = 0; k < 3; ++k) ctr[e][k] = (float)((w.q[k] >> (8 * e)) & 255u) + (float)((w.q[3 + k] >> (8 * e)) & 255u);
                nv = e + 1;
            }
            int ax = 0;
            if (order == 2) {
                float spread[3];
                const float sc3[3] = { w.scale_x, w.scale_y, w.scale_z };
                for (int k = 0; k < 3; ++k) { float lo = 1e30f, hi = -1e30f; for (int e = 0; e < nv; ++e) { lo = fminf(lo, ctr[e][k]); hi = fmaxf(hi, ctr[e][k]); } spread[k] = (hi - lo) * sc3[k]; }
                ax = spread[1] > spread[0] ? 1 : 0; if (spread[2] > spread[ax]) ax = 2;
            }
            for (int e = 0; e < 4; ++e) {
                if (w.child[e] == EMPTY) continue;
                float a = -std::numeric_limits<float>::infinity(), b = std::numeric_limits<float>::infinity();
                for (int k = 0; k < 3; ++k) {
                    const float sc = k == 0 ? w.scale_x : (k == 1 ? w.scale_y : w.scale_z);
                    const float lo = pt_grid_plane((w.q[k] >> (8 * e)) & 255u, sc, w.origin[k]);
                    const float hi = pt_grid_plane((w.q[3 + k] >> (8 * e)) & 255u, sc, w.origin[k]);
                    const float x = __builtin_fmaf(lo, inv[k], -oo[k]), y = __builtin_fmaf(hi, inv[k], -oo[k]);
                    a = fmaxf(a, fminf(x, y)); b = fminf(b, fmaxf(x, y));
                }
                if (fmaxf(a, tmin) <= fminf(b, (float)best)) {
                    if (order == 1) a = (float)e;
                    else if (order == 2) a = d[ax] < 0 ? -ctr[e][ax] : ctr[e][ax];
                    else if (order == 3) a = (d[0] < 0 ? -ctr[e][0] : ctr[e][0]) * w.scale_x + (d[1] < 0 ? -ctr[e][1] : ctr[e][1]) * w.scale_y + (d[2] < 0 ? -ctr[e][2] : ctr[e][2]) * w.scale_z;
                    tn[nh] = a; rf[nh] = w.child[e]; ++nh;
                }
            }
            if (order == 5) {         // the nearest first, the others as stored
                int best_i = 0;
                for (int i = 1; i < nh; ++i) if (tn[i] < tn[best_i]) best_i = i;
                const float t0 = tn[best_i]; const int32_t r0 = rf[best_i];
                for (int i = best_i; i > 0; --i) { tn[i] = tn[i - 1]; rf[i] = rf[i - 1]; }
                if (nh) { tn[0] = t0; rf[0] = r0; }
            } else
            for (int i = 1; i < nh; ++i) for (int j = i; j > 0 && tn[j] < tn[j - 1]; --j) { std::swap(tn[j], tn[j - 1]); std::swap(rf[j], rf[j - 1]); }
            for (int i = nh - 1; i >= 0; --i) stack.push_back(rf[i]);
            deepest = stack.size() > deepest ? stack.size() : deepest;
            // stats[3 + k]: node steps that leave k or more entries below the child taken next (k = 1..20)
            if (order == 0) for (size_t k = 1; k <= 20 && k + 1 <= stack.size(); ++k) ++stats[3 + k];
        }
        if (deepest > stats[2]) stats[2] = deepest;
        if (t_out) t_out[r] = (float)best;
    }
}

}
