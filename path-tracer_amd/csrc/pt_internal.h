// pt_internal.h -- structures shared by the translation units of libpt_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/pt_hip.h"

void pt_set_error(const char* fmt, ...);

#define PT_HIP_CHECK(expr)                                                                   \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            (void)hipGetLastError();   /* reported here: a later hipGetLastError() must not see it again */ \
            pt_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return e_ == hipErrorOutOfMemory ? PT_ERROR_OUT_OF_MEMORY : PT_ERROR_HIP;        \
        }                                                                                    \
    } while (0)

#define PT_REQUIRE(cond, msg)                      \
    do {                                           \
        if (!(cond)) {                             \
            pt_set_error("%s: %s", __func__, msg); \
            return PT_ERROR_INVALID_ARGUMENT;      \
        }                                          \
    } while (0)

// ---- device-side records -------------------------------------------------------------------
// Traversal view of one TLAS instance: six float4.
//   [0..2] world->object rows; [3] = (box lo.xyz, delta); [4] = (box hi.xyz, bits(custom index));
//   [5] = (bits(root node index of the BLAS inside the TLAS pool), bits(instance index), -, -)
// The TLAS keeps them in the order of its leaves: record k belongs to leaf slot k.
struct PtDevInstance { float4 q[6]; };
// Shading view of one instance: the bindless chain blasInstances[custom] -> geometryNodes[first + 0] ->
// materials[materialIndex] -> textures[albedoMapIndex] (closest_hit.rchit:57-59,75-79) is constant per
// instance, so it is resolved once on the host into this 160-byte record.
struct PtDevShadeInstance {
    float4 o2w[3];
    float4 w2o[3];
    float4 ptrs;      // pointer lo/hi to the instance's triangle records (three 32-byte vertices per primitive), -, -
    float4 albedo;    // albedoFactor.rgb, bits(useAlbedoMap)
    float4 emissive;  // emissiveFactor.rgb, -
    float4 tex;       // texel pointer lo/hi, bits(width), bits(height)
};

struct PtBlas {
    PtContext* ctx;
    const uint8_t* vertices; uint32_t stride; uint32_t max_vertex;
    const uint32_t* indices; uint32_t prim_count; uint32_t max_leaf;
    float delta; float root_lo[3], root_hi[3];
    uint32_t node_count;
    PtBvhNode* nodes;        // device, node_count
    float4* tris;            // device, 3 float4 per sorted slot
    uint32_t* leaf_prims;    // device, prim ids in sorted order
    uint64_t* sorted_keys;   // device
    PtBvhNode* trav_nodes;   // device: binned-SAH binary tree the traversal layout is derived from (null: the Morton tree is)
    uint32_t trav_node_count;
    uint32_t* trav_order;    // device: primitive at each leaf position of that tree (tris follow it)
    void* wide;              // device: 64-byte 4-wide quantised nodes of this BLAS, references local (root 0, triangle 0)
    uint32_t wide_count, wide_depth;
    void* block;             // device: the one allocation trav_nodes, wide, tris and trav_order are parts of (null: each has its own)
    float build_ms;
};

#define PT_ROOT_ITEMS_MAX 8          // largest TLAS whose rays may start with the instances instead of the root node

struct PtTlas {
    PtContext* ctx;
    uint32_t n;
    uint32_t node_count;
    PtBvhNode* nodes;               // device: TLAS nodes alone (what pt_tlas_download returns)
    void* pool_nodes;               // device: 64-byte 4-wide quantised nodes, TLAS first then every distinct BLAS, references absolute
    int n_root;                     // instances a ray starts with instead of the TLAS root (see pt_ray_start), or 0
    void* pool_nodes_f32;           // device: the same nodes as 128-byte PtWideF (float planes), what k_primary traverses
    float4* pool_tris;              // device: triangle records of every distinct BLAS, 3 float4 each
    uint32_t pool_node_count;
    uint32_t wide_count, wide_depth;  // the TLAS's own wide nodes: the first wide_count of the pool
    uint32_t* leaf_inst;            // device, instance index per sorted slot (Morton tree)
    PtBvhNode* trav_nodes;          // device: full-sweep SAH tree over the instance boxes the wide nodes are derived from (null: the Morton tree is)
    uint32_t trav_node_count;
    std::vector<uint32_t> trav_order;   // instance at each leaf position of that tree
    PtDevInstance* inst;            // device, n, in leaf-slot order
    PtDevShadeInstance* shade_inst; // device, n; rebuilt when the context's tables change
    uint64_t shade_version;         // ctx->tables_version the records were built from
    std::vector<float> host_o2w;    // n x 12
    std::vector<uint32_t> host_custom;
    std::vector<uint32_t> host_prim_count;   // primitives of each instance's BLAS
    // Shading input per primitive, gathered through the geometry node's own buffers when the tables
    // are bound: instances with the same buffers share one array.
    struct TriRecords { uint64_t vertices, indices; uint32_t prims; float4* dev; };
    std::vector<TriRecords> tri_records;
    std::vector<float> host_boxes;  // n x 6
    std::vector<float> host_w2o;    // n x 12
    float root_lo[3], root_hi[3];
    float build_ms;
};

// One frame in flight: its own stream and path queues, so that the thin late bounces of one frame
// overlap the wide early bounces of the next.  The reference cycles its per-frame command buffers
// and fences the same way (source/renderer.cpp:62, MAX_FRAMES_IN_FLIGHT in include/vk_common.hpp:6).
// Device scratch of the builders, kept between builds: chunks of device memory, filled in order.
struct PtArena {
    std::vector<std::pair<char*, size_t>> chunks;
    size_t cur = 0, used = 0;      // chunk being filled and bytes used in it
};

struct PtLane {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;         // recorded after the latest kernel enqueued on this lane
    hipEvent_t accumulated = nullptr;  // recorded after the latest k_accumulate on this lane
    bool pending = false;              // work enqueued since the last join
    uint64_t paths_cap = 0;            // slots per stream = kShards * shard_cap
    uint32_t shard_cap = 0;            // slots per shard (multiple of 64)
    uint64_t shard_cap_wanted = 0;     // largest capacity ever asked for
    bool rays_only = false;            // only q[0][0] and q[0][1] are allocated (pt_intersect)
    float4* q[2][4] = { { nullptr, nullptr, nullptr, nullptr }, { nullptr, nullptr, nullptr, nullptr } };
    PtHit* hits = nullptr;
    float4* path_result = nullptr;     // per path: final hitValue (xyz)
    uint32_t* counters = nullptr;      // queue sizes per depth (+ work counters)
    int* spill = nullptr;              // traversal stack overflow, PT_SPILL_STACK ints per persistent thread
};

struct PtWorkspace {                   // shared by all lanes
    float* frame_result = nullptr;     // per pixel x 3
    uint64_t frame_pixels_cap = 0;
    uint64_t* stats = nullptr;         // PT_MAX_DEPTH rays-per-depth + paths
    uint64_t spill_threads = 0;
};

// Per-lane traversal stack capacity: entries in LDS, then entries in a per-thread global slice.
#ifndef PT_LDS_STACK
#define PT_LDS_STACK 16
#endif
#ifndef PT_SPILL_STACK
#define PT_SPILL_STACK 112
#endif
// entries the shallowest LDS stack of any traversal kernel holds (the experimental two-rays-per-lane kernel keeps 10 per ray)
#ifdef PT_EXPERIMENTAL
#define PT_MIN_LDS_STACK (PT_LDS_STACK < 10 ? PT_LDS_STACK : 10)
#else
#define PT_MIN_LDS_STACK PT_LDS_STACK
#endif

struct PtContext {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cus = 0;
    uint64_t max_paths = 1ull << 27;
    int traversal_mode = 2;   // 0: one batch of 64 rays per wave at a time; 1: wave-scheduled with ray replacement; 2: chosen per level; (PT_HIP_TRAVERSAL=plain|sched|auto); builds with PT_EXPERIMENTAL only: 3 wave-scheduled with two rays per lane (pair), 4 / 5 rays pooled in LDS everywhere / above pool_min_rays (pool, autopool)
    uint32_t pool_min_rays = 1200000;    // mode 4 (PT_HIP_TRAVERSAL=pool): levels with at least this many rays are traversed from LDS ray pools (PT_HIP_POOL_MIN_RAYS)
    uint32_t sched_min_rays = 1200000;   // mode 2: levels with at least this many rays are wave-scheduled (PT_HIP_SCHED_MIN_RAYS)
    // tables
    PtMaterial* materials = nullptr; uint32_t n_materials = 0;
    PtGeometryNode* geometry_nodes = nullptr; uint32_t n_geometry_nodes = 0;
    PtBLASInstance* blas_instances = nullptr; uint32_t n_blas_instances = 0;
    PtTextureDesc* textures = nullptr; uint32_t n_textures = 0;
    std::vector<PtMaterial> h_materials; std::vector<PtGeometryNode> h_geometry_nodes;
    std::vector<PtBLASInstance> h_blas_instances; std::vector<PtTextureDesc> h_textures;
    uint64_t tables_version = 0;
    PtWorkspace ws;
    std::vector<PtLane> lanes;                 // frames in flight (PT_HIP_FRAMES_IN_FLIGHT, pt_ctx_set_frames_in_flight)
    uint32_t next_lane = 0;
    PtLane* last_accumulate = nullptr;         // lane whose k_accumulate every later one has to follow
    hipEvent_t start_ev = nullptr;             // marks the caller's stream at the start of a unit of work
    uint32_t last_w = 0, last_h = 0;
    bool share_primary = false;                // bounce 0 is traced once per pixel and shared by the pixel's samples (pt_ctx_set_share_primary)
    // intersect-kernel timing (pt_ctx_set_profiling)
    bool profiling = false;
    std::vector<hipEvent_t> ev_pool;           // pairs: start, stop
    size_t ev_used = 0;
    std::vector<uint8_t> ev_primary;           // per pair: the launch was bounce 0 (k_primary)
    PtArena build_arena;
    uint32_t* pinned_word = nullptr;           // pinned host word for single-value readbacks
    uint32_t* mailbox = nullptr;               // mapped, coherent host words that kernels write and the host polls (per-level totals of the builders)
    uint32_t* mailbox_dev = nullptr;           // ... as the device addresses them
    std::vector<uint32_t> ev_depth;            // depth of each recorded pair
    uint64_t stats_mark[PT_MAX_DEPTH] = {};    // rays_per_depth at the last pt_trace_profile
    unsigned char* na_scratch = nullptr;       // pt_intersect_noalloc: device address of this GPU's fixed workspace ...
    int na_blocks = 0;                         // ... and the grid of its traversal launch
};

// pt_capi.hip: frames in flight
int pt_lanes_resize(PtContext* ctx, int n);                 // waits for everything, then recreates the lanes
int pt_lane_begin(PtContext* ctx, PtLane** out);            // next lane, ordered after the caller's stream
int pt_lanes_join(PtContext* ctx);                          // the caller's stream waits for every lane (no host wait)
int pt_lanes_sync(PtContext* ctx);                          // the host waits for every lane and the caller's stream

// pt_trace.hip: looks up the device address of the fixed workspace of pt_intersect_noalloc (once per device)
int pt_noalloc_setup(PtContext* ctx);

// pt_build.hip
int pt_build_wide(PtContext* ctx, const PtBvhNode* nodes_dev, uint32_t node_count, void** wide_out, uint32_t* count_out, uint32_t* depth_out);
int pt_build_lbvh(PtContext* ctx, const float* prim_boxes_dev /* n x 6: lo, hi */, uint32_t n, uint32_t max_leaf,
                  PtBvhNode** nodes_out, uint32_t* node_count_out, uint32_t** leaf_prims_out, uint64_t** sorted_keys_out);
