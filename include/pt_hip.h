/*
 * pt_hip.h -- C ABI of libpt_hip.so: the MI355X (gfx950) replacement for the Vulkan
 * ray-tracing entry points the reference's hot path goes through.
 *
 * The reference has no FFI for this path: its C++ classes call Vulkan directly.  Each entry
 * point below names the Vulkan call site(s) (reference file:line) it replaces.  All functions
 * return 0 on success and a non-zero PtStatus otherwise; pt_last_error() returns a thread-local
 * description (the reference logs and abort()s instead, source/vk_common.cpp:5-15 -- the C++
 * wrappers may do the same on a non-zero status).
 *
 * Conventions: plain pointers and sizes only; "_dev" arguments are device pointers valid on the
 * context's GPU; everything else is host memory.  One host thread per context; calls on a context
 * are asynchronous on its HIP stream unless stated, pt_sync() waits for them.
 */
#ifndef PT_HIP_H
#define PT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum PtStatus {
    PT_OK = 0,
    PT_ERROR_INVALID_ARGUMENT = 1,
    PT_ERROR_HIP = 2,
    PT_ERROR_OUT_OF_MEMORY = 3,
    PT_ERROR_LIMIT = 4
} PtStatus;

typedef struct PtContext PtContext;
typedef struct PtBlas PtBlas;
typedef struct PtTlas PtTlas;

/* ---- device-visible table layouts (shaders/bindless.glsl:3-51,
 *      include/resources/gpu_resources.hpp:157-205) */
typedef struct PtMaterial {              /* std140, 96 bytes */
    float    albedoFactor[4];
    float    metallicFactor, roughnessFactor, normalScale, occlusionStrength;
    float    emissiveFactor[3];
    int32_t  useEmissiveMap;
    int32_t  useAlbedoMap, useMRMap, useNormalMap, useOcclusionMap;
    uint32_t albedoMapIndex, mrMapIndex, normalMapIndex, occlusionMapIndex;
    uint32_t emissiveMapIndex;
    uint32_t pad[3];
} PtMaterial;

typedef struct PtGeometryNode {          /* 32 bytes */
    uint64_t vertexBufferDeviceAddress;  /* interleaved 32-byte vertices {pos, normal, uv} */
    uint64_t indexBufferDeviceAddress;   /* u32 indices, already offset to the mesh's first index */
    uint32_t materialIndex;
    uint32_t pad[3];
} PtGeometryNode;

typedef struct PtBLASInstance { uint32_t firstGeometryIndex; } PtBLASInstance;

/* VkAccelerationStructureInstanceKHR (64 bytes) as filled by
 * source/top_level_acceleration_structure.cpp:25-36; accelerationStructureReference holds the
 * PtBlas* returned by pt_blas_build. */
typedef struct PtInstance {
    float    transform[12];              /* row-major 3x4 object->world (source/vk_common.cpp:205-212) */
    uint32_t instanceCustomIndexAndMask; /* [23:0] custom index, [31:24] mask */
    uint32_t sbtOffsetAndFlags;          /* [23:0] SBT record offset, [31:24] VkGeometryInstanceFlags */
    uint64_t accelerationStructureReference;
} PtInstance;

typedef struct PtTextureDesc { uint64_t rgba8_dev; uint32_t width, height; } PtTextureDesc;

typedef struct PtHit { uint32_t instance, primitive; float u, v; } PtHit;   /* instance = 0xFFFFFFFF: miss */

/* 64-byte BVH node: both child boxes and both child references.
 * child >= 0: inner node index; child < 0: leaf, ~child = first << 3 | (count - 1), positions in
 * the sorted primitive order; INT32_MIN: empty slot. */
typedef struct PtBvhNode {
    float   lo0[3], hi0[3], lo1[3], hi1[3];
    int32_t c0, c1;
    uint32_t aux0, aux1;
} PtBvhNode;

typedef struct PtAsInfo {
    uint32_t node_count, prim_count;
    float    root_lo[3], root_hi[3];
    float    delta;                      /* box padding of this structure */
    float    build_ms;                   /* device time of the build */
} PtAsInfo;

#define PT_MAX_DEPTH 64
typedef struct PtTraceStats {
    uint64_t rays_per_depth[PT_MAX_DEPTH]; /* rays traced at each bounce since pt_trace_stats_reset */
    uint64_t rays, paths;
} PtTraceStats;

enum { PT_SEED_SAMPLE_INDEX = 0,  /* seed = TEA(pixel, frame * spp + i): deterministic */
       PT_SEED_CLOCK = 1 };       /* seed = TEA(pixel, clock): what ray_gen.rgen:40 does */
enum { PT_ACCUM_SUM_F32 = 0,      /* float4 per pixel: rgb += frame result, a += 1 */
       PT_ACCUM_MEAN_F32 = 1,     /* float4 per pixel: literal mix() of ray_gen.rgen:77-87 */
       PT_ACCUM_MEAN_UNORM8 = 2 };/* 4 x u8 per pixel: literal mix() on an 8-bit UNORM image */

const char* pt_last_error(void);

/* Replaces VulkanContext device/queue selection (source/vulkan_context.cpp:99-113): binds a GPU
 * and a stream.  hip_stream = NULL: the library creates a non-blocking stream of its own, which is
 * NOT ordered with the null stream or any stream of the caller (use pt_sync / pt_trace_fence and
 * your own synchronisation around shared buffers).  Otherwise hip_stream is the hipStream_t to
 * enqueue on, and everything this header says about "the context's stream" holds for it.  The null
 * (legacy default) stream has the handle 0 and therefore cannot be passed as such: pass
 * hipStreamLegacy or hipStreamPerThread for it, or a stream of your own (a framework's current
 * stream may be the null stream: PyTorch's is unless a torch.cuda.Stream has been made current). */
int pt_ctx_create(int device_id, void* hip_stream, PtContext** out);
int pt_ctx_destroy(PtContext* ctx);
/* Waits on the host for everything the context has enqueued, frames in flight included. */
int pt_sync(PtContext* ctx);

/* Replaces the per-frame command buffers and fences the reference cycles through
 * (source/renderer.cpp:52-62, MAX_FRAMES_IN_FLIGHT = 3 in include/vk_common.hpp:6): pt_trace puts
 * every frame (every pixel chunk of it) on the next of n internal streams, each with its own path
 * queues, and returns without waiting, so the thin late bounces of one frame overlap the wide early
 * bounces of the next.  Work the caller enqueued on its own stream before a pt_trace call is
 * waited for by that frame; updates of the accumulation image happen in submission order.
 * Default 3 (environment: PT_HIP_FRAMES_IN_FLIGHT); 1 restores strictly serial frames.
 * Changing the number waits for all frames and releases their queues. */
int pt_ctx_set_frames_in_flight(PtContext* ctx, int n);
/* Vulkan's vkWaitForFences(_inFlightFences) of source/renderer.cpp:64-65 as a stream operation:
 * makes the context's stream wait for all frames in flight without blocking the host.  Call it
 * before enqueuing your own work that reads the accumulation image on that stream; every entry
 * point of this library that reads results (pt_accum_resolve, pt_trace_last_result,
 * pt_trace_stats, pt_trace_profile, pt_buffer_download, pt_intersect's outputs) does so itself. */
int pt_trace_fence(PtContext* ctx);

/* Replaces the staging-buffer uploads of source/model_loader.cpp:307-357 and
 * source/resources/gpu_resources.cpp:207-271: hipMalloc + copy.  Synchronous. */
int pt_buffer_upload(PtContext* ctx, const void* host, size_t bytes, void** dev_out);
int pt_buffer_download(PtContext* ctx, const void* dev, size_t bytes, void* host_out);
int pt_buffer_free(PtContext* ctx, void* dev);

/* Replaces getAccelerationStructureBuildSizesKHR + createAccelerationStructureKHR +
 * cmdBuildAccelerationStructuresKHR for one triangle geometry
 * (source/bottom_level_acceleration_structure.cpp:34-78, inputs source/renderer.cpp:417-455):
 * R32G32B32 positions at vertex_stride, u32 indices, opaque, build mode only.
 * max_leaf = 0 selects the default (4).  Blocks until built, like the reference's
 * SingleTimeCommands::SubmitAndWait.  The BLAS keeps references to the caller's buffers. */
int pt_blas_build(PtContext* ctx, const void* vertices_dev, uint32_t vertex_stride, uint32_t max_vertex,
                  const uint32_t* indices_dev, uint32_t prim_count, uint32_t max_leaf, PtBlas** out);
int pt_blas_destroy(PtBlas* blas);   /* destroyAccelerationStructureKHR, bottom_level_...cpp:19 */
/* node_count is that of the Morton-order tree below: 0 until that tree exists. */
int pt_blas_info(const PtBlas* blas, PtAsInfo* out);
/* Copies the Morton-order binary tree to the host (any pointer may be NULL): nodes[node_count],
 * leaf_prims[prim_count] (primitive ids in sorted order), sorted_keys[prim_count].  The kernels traverse
 * the binned-SAH tree (pt_blas_traversal_download), so the Morton tree is built by the first call of this
 * function, from the vertex and index buffers given to pt_blas_build; call it with three NULLs to have
 * the tree built and pt_blas_info report its node_count. */
int pt_blas_download(const PtBlas* blas, PtBvhNode* nodes, uint32_t* leaf_prims, uint64_t* sorted_keys);
/* The layout the kernels traverse, derived on the GPU from a binary tree (pt_blas_traversal_download
 * below; for a TLAS its Morton tree): 64-byte nodes
 * with four children whose boxes are quantised to 8 bits on a per-node power-of-two grid
 * { float origin[3]; float step_x; int32 child[4]; uint32 q[6] (lo.x lo.y lo.z hi.x hi.y hi.z, byte k
 * = child k); float step_y, step_z }, numbered breadth first, references local to the structure.
 * nodes64 may be NULL to query the count and depth only.  A driver keeps its traversal format
 * private (the reference only sees buffer sizes, bottom_level_...cpp:95-110); it is exposed here
 * so that it can be checked. */
int pt_blas_wide_download(const PtBlas* blas, void* nodes64, uint32_t capacity, uint32_t* count_out, uint32_t* depth_out);
/* The binary tree the wide nodes are derived from, in the node format of pt_blas_download, and the
 * primitive at each of its leaf positions.  By default this is a top-down binned-SAH tree built on the
 * GPU next to the Morton tree (the reference asks its driver for ePreferFastTrace,
 * bottom_level_...cpp:58); with PT_HIP_TRAVERSAL_TREE=lbvh it is the Morton tree itself.  The closest
 * hit does not depend on the tree.  nodes may be NULL to query the count. */
int pt_blas_traversal_download(const PtBlas* blas, PtBvhNode* nodes, uint32_t capacity, uint32_t* leaf_prims, uint32_t* count_out);

/* Replaces the TLAS build of source/top_level_acceleration_structure.cpp:19-113: n instance
 * records in host memory.  Blocks until built.  The traversal pool of one TLAS (its own 4-wide nodes and
 * those of every distinct BLAS under it) is addressed with 32-bit byte offsets into 128-byte records:
 * PT_ERROR_LIMIT when it would hold PT_MAX_POOL_NODES nodes or more, or 2^28 triangles or more. */
#define PT_MAX_POOL_NODES (1u << 25)
int pt_tlas_build(PtContext* ctx, const PtInstance* instances, uint32_t n, PtTlas** out);
int pt_tlas_destroy(PtTlas* tlas);   /* top_level_...cpp:16 */
int pt_tlas_info(const PtTlas* tlas, PtAsInfo* out);
int pt_tlas_download(const PtTlas* tlas, PtBvhNode* nodes, uint32_t* leaf_instances,
                     float* instance_boxes /* n x 6 */, float* instance_w2o /* n x 12 */);
/* The binary tree the TLAS's wide nodes are derived from and the instance at each of its leaf positions: the
 * cheaper, in the surface-area model, of the Morton tree of pt_tlas_download and a full-sweep SAH tree over the
 * instance boxes (the driver is asked for ePreferFastTrace, top_level_acceleration_structure.cpp:75);
 * PT_HIP_TLAS_TREE=lbvh|sweep forces one.  The closest hit does not depend on the tree.  nodes may be NULL to
 * query the count. */
int pt_tlas_traversal_download(const PtTlas* tlas, PtBvhNode* nodes, uint32_t capacity, uint32_t* leaf_instances, uint32_t* count_out);
int pt_tlas_wide_download(const PtTlas* tlas, void* nodes64, uint32_t capacity, uint32_t* count_out, uint32_t* depth_out);

/* Replaces BindlessResources::UpdateDescriptorSet (source/resources/bindless_resources.cpp:70-76,
 * bindings 0-3 of set 0, :250-274): copies the four tables to the GPU.  Table sizes are capped at
 * 1024 entries like the reference (include/resources/bindless_resources.hpp:65).
 * The vertices a geometry node addresses are read when a TLAS is next traced (per-primitive copies
 * for shading); call again after changing vertex or index data in place. */
int pt_scene_set_tables(PtContext* ctx,
                        const PtMaterial* materials, uint32_t n_materials,
                        const PtGeometryNode* geometry_nodes, uint32_t n_geometry_nodes,
                        const PtBLASInstance* blas_instances, uint32_t n_blas_instances,
                        const PtTextureDesc* textures, uint32_t n_textures);

/* traceRayEXT on caller-supplied rays (shaders/ray_gen.rgen:54-65 without the shaders): closest
 * hit over the TLAS, opaque, no culling, tmin < t < tmax.  origins/dirs are n x 3 floats;
 * t_dev may be NULL. */
int pt_intersect(PtContext* ctx, const PtTlas* tlas, uint32_t n, const float* origins_dev, const float* dirs_dev,
                 float tmin, float tmax, PtHit* hits_dev, float* t_dev);

/* pt_intersect for callers that cannot allocate device memory: code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  Has no counterpart in the reference.  Same
 * arguments and results as pt_intersect; the ray queues, counters and stack slices come from a fixed
 * array that every GPU the library runs on holds until the process ends (89 MiB), so n is limited to
 * 1,048,576 rays (PT_ERROR_LIMIT above that) and the call neither allocates nor frees.  It runs on
 * the context's stream.  Every call on one GPU shares that GPU's array: calls that overlap on one GPU
 * (other streams, other contexts, other threads) race; calls on one stream follow each other. */
#define PT_INTERSECT_NOALLOC_MAX_RAYS (1u << 20)
int pt_intersect_noalloc(PtContext* ctx, const PtTlas* tlas, uint32_t n, const float* origins_dev, const float* dirs_dev,
                         float tmin, float tmax, PtHit* hits_dev, float* t_dev);

/* Replaces cmdTraceRaysKHR(width, height, 1) with the ray_gen / closest_hit / miss pipeline
 * (source/renderer.cpp:109-122): one frame of samples_per_frame paths per pixel, at most
 * max_depth rays per path, accumulated into accum_dev according to accum_mode.
 * camera = CameraUniformData (include/renderer.hpp:31-35): viewInverse then projInverse,
 * column-major.  frame is the push constant (include/renderer.hpp:37-40).
 * The reference's literals are samples_per_frame 25, max_depth 10 (shaders/ray_gen.rgen:34,52). */
int pt_trace(PtContext* ctx, const PtTlas* tlas, const float camera[32], uint32_t width, uint32_t height,
             uint32_t frame, uint32_t samples_per_frame, uint32_t max_depth,
             uint32_t seed_mode, uint32_t accum_mode, void* accum_dev);

/* Option of pt_trace, off by default.  The reference's primary ray does not depend on the sample
 * (shaders/ray_gen.rgen:21-27: no jitter; the sample loop :38-47 starts every path with the same origin and
 * direction), so the traceRayEXT of bounce 0 (:54-65) returns the same hit for all samples of a pixel.  With the
 * option on it is executed once per pixel and its hit is handed to the pixel's samples, which are shaded one by
 * one as before: images are bit-identical, bounce 0 costs one traversal per pixel instead of one per sample, and
 * pt_trace_stats counts the rays that were traversed (rays_per_depth[0] = pixels, not paths). */
int pt_ctx_set_share_primary(PtContext* ctx, int enabled);

/* Per-pixel value of `result` (shaders/ray_gen.rgen:74) of the last pt_trace: width*height*3
 * floats to host memory.  Synchronous; for tests. */
int pt_trace_last_result(PtContext* ctx, float* result_host);
int pt_trace_stats(PtContext* ctx, PtTraceStats* out);       /* synchronous */
int pt_trace_stats_reset(PtContext* ctx);
/* Timing of the intersect kernel with HIP events on the streams it is launched on (off by
 * default).  While enabled, every intersect launch of pt_trace is bracketed by an event pair;
 * pt_trace_profile waits for all frames, adds up the pairs recorded since the last call and
 * returns them.  With several frames in flight the launches of different frames overlap, so the
 * sum is kernel time, not elapsed time.  Has no counterpart in the reference (it has no timers). */
typedef struct PtTraceProfile {
    uint64_t intersect_launches;   /* k_intersect: bounces after the first */
    uint64_t intersect_rays;
    double   intersect_ms;
    uint64_t primary_launches;     /* k_primary: bounce 0, traced and shaded in one kernel */
    uint64_t primary_rays;
    double   primary_ms;
} PtTraceProfile;
int pt_ctx_set_profiling(PtContext* ctx, int enabled);
int pt_trace_profile(PtContext* ctx, PtTraceProfile* out);

/* Upper bound on paths resident at once per frame in flight, in [64, 2^31) (default 1<<27, about
 * 17 GB of queues when reached; queues are sized by the largest chunk actually traced); a larger
 * frame is rendered in pixel chunks. */
int pt_ctx_set_max_paths(PtContext* ctx, uint64_t max_paths);

/* Replaces the render-target -> swapchain blit (source/renderer.cpp:124-133): float sum buffer
 * -> mean -> clamp -> RGBA8 (bgra != 0: B8G8R8A8, the preferred swapchain format,
 * source/swap_chain.cpp:122-133). */
int pt_accum_resolve(PtContext* ctx, const float* accum_sum_dev, uint32_t n_pixels, uint8_t* rgba8_dev, int bgra);

/* Multi-GPU in one process: sums the float accumulation images (PT_ACCUM_SUM_F32, n_pixels float4
 * each) of n contexts, one per GPU, so that every image holds the total.  The reference renders on one
 * device (source/vulkan_context.cpp:218-286); this is the exchange step of samples-per-pixel
 * sharding: context r traces frames r, r + n, r + 2n, ... into accum_sum_dev[r], then one call here.
 * It goes over xGMI peer access directly, every link at once: GPU r sums slice r of all images in the
 * order of the contexts (the result is the same bits on every GPU and from run to run) and every GPU
 * then copies the other slices from their owners.  Asynchronous: each context's stream waits for its
 * frames in flight and for the other contexts; pt_sync a context before reading its image on the host.
 * n <= 16; contexts may share a device (rehearsal on one GPU); devices must have peer access. */
int pt_accum_allreduce(PtContext* const* ctxs, void* const* accum_sum_dev, uint32_t n, uint32_t n_pixels);

/* No counterpart in the reference: a self-test of the library's arithmetic on the GPU in use.  The tracing kernels
 * compute 1/x, sqrt(x), x/pi and byte/255 with the hardware approximations and one residual correction instead of
 * the IEEE expansions, and divide path indices by samples-per-pixel and image width with a multiplier; results
 * must have the bits of the correctly rounded operations (what a CPU computes for the GLSL's `/` and sqrt).
 * This call compares them over every binary32 pattern (every byte; eighteen divisors and every third index below
 * 2^31) and returns the number of differing results: mismatches[0..4] = reciprocal, square root, x/pi, byte/255,
 * integer division.  All five must be zero.  Synchronous, about a second. */
int pt_selftest_arith(PtContext* ctx, uint64_t mismatches[5]);

#ifdef __cplusplus
}
#endif
#endif
