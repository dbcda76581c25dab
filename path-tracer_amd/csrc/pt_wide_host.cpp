// pt_wide_host.cpp -- C entry points over the host-side wide-node construction, for checking it
// without a GPU: the collapse itself and a plain enumeration of the leaves a ray reaches.
#include "pt_wide.h"
#include "pt_sah.h"

namespace {
const float kDirEps = 8.271806125530277e-25f;   // 2^-80
inline float safe_rcp(float d) { float s = fabsf(d) > kDirEps ? d : copysignf(kDirEps, d); return 1.0f / s; }
}

extern "C" {

// Collapses a binary BVH into wide nodes; returns the number of wide nodes (0 if `cap` is too small).
uint32_t pt_wide_collapse_host(const PtBvhNode* nodes, uint32_t node_count, uint32_t tri_base,
                               void* out, uint32_t cap, uint32_t* depth)
{
    std::vector<PtBvhNode> n2(nodes, nodes + node_count);
    std::vector<PtWide4> w;
    uint32_t d = pt_collapse_to_wide(n2, tri_base, w);
    if (depth) *depth = d;
    if (w.size() > cap) return 0;
    memcpy(out, w.data(), sizeof(PtWide4) * w.size());
    return (uint32_t)w.size();
}

// Host twin of the GPU binned-SAH builder: boxes n x 6, nodes_out capacity n, order_out n; returns the node count.
uint32_t pt_sah_build_host_c(const float* boxes, uint32_t n, uint32_t max_leaf, const float* root_lo, const float* root_hi,
                             PtBvhNode* nodes_out, uint32_t* order_out)
{
    std::vector<PtBvhNode> nodes; std::vector<uint32_t> ids;
    pt_sah_build_host(boxes, n, max_leaf, root_lo, root_hi, nodes, ids);
    memcpy(nodes_out, nodes.data(), sizeof(PtBvhNode) * nodes.size());
    memcpy(order_out, ids.data(), sizeof(uint32_t) * n);
    return (uint32_t)nodes.size();
}

// Full-sweep SAH tree over n boxes (the TLAS traversal tree): nodes_out capacity n, order_out n; returns the node count.
uint32_t pt_sweep_sah_build_host_c(const float* boxes, uint32_t n, PtBvhNode* nodes_out, uint32_t* order_out)
{
    std::vector<PtBvhNode> nodes; std::vector<uint32_t> ids;
    pt_sweep_sah_build_host(boxes, n, nodes, ids);
    memcpy(nodes_out, nodes.data(), sizeof(PtBvhNode) * nodes.size());
    memcpy(order_out, ids.data(), sizeof(uint32_t) * n);
    return (uint32_t)nodes.size();
}

double pt_tree_cost_host_c(const PtBvhNode* nodes, uint32_t n) { return pt_tree_cost_host(nodes, n); }

// Writes the leaf codes (first << 3 | count - 1) of every leaf whose chain of dequantised boxes the
// ray interval [tlo, thi] overlaps, in traversal order without any pruning by found hits.
uint32_t pt_wide_candidates_host(const void* wide, uint32_t root, const float o[3], const float d[3],
                                 float tlo, float thi, uint32_t* out_codes, uint32_t cap)
{
    const PtWide4* W = (const PtWide4*)wide;
    float inv[3], oo[3];
    for (int k = 0; k < 3; ++k) { inv[k] = safe_rcp(d[k]); oo[k] = o[k] * inv[k]; }
    std::vector<int32_t> stack { (int32_t)root };
    uint32_t n = 0;
    const int32_t EMPTY = std::numeric_limits<int32_t>::min();
    while (!stack.empty()) {
        const int32_t ref = stack.back(); stack.pop_back();
        if (ref < 0) { if (n < cap) out_codes[n] = ~(uint32_t)ref; ++n; continue; }
        const PtWide4& w = W[ref];
        for (int e = 0; e < 4; ++e) {
            if (w.child[e] == EMPTY) continue;
            float tn = -std::numeric_limits<float>::infinity(), tf = std::numeric_limits<float>::infinity();
            for (int k = 0; k < 3; ++k) {
                const float sc = k == 0 ? w.scale_x : (k == 1 ? w.scale_y : w.scale_z);
                const float lo = pt_grid_plane((w.q[k] >> (8 * e)) & 255u, sc, w.origin[k]);
                const float hi = pt_grid_plane((w.q[3 + k] >> (8 * e)) & 255u, sc, w.origin[k]);
                const float a = __builtin_fmaf(lo, inv[k], -oo[k]), b = __builtin_fmaf(hi, inv[k], -oo[k]);
                tn = fmaxf(tn, fminf(a, b)); tf = fminf(tf, fmaxf(a, b));
            }
            if (fmaxf(tn, tlo) <= fminf(tf, thi)) stack.push_back(w.child[e]);
        }
    }
    return n;
}


}
