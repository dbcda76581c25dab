// pt_capi.hip -- context, buffers, tables and error reporting of the C ABI (include/pt_hip.h).
#include "pt_internal.h"
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

static thread_local char g_error[512] = "";

void pt_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
}

// Allocates and fills the new table first: a failed allocation or copy leaves the old table bound.
template <typename T>
static int replace_table(T** dst, uint32_t* n_dst, const T* src, uint32_t n)
{
    T* fresh = nullptr;
    PT_HIP_CHECK(hipMalloc((void**)&fresh, sizeof(T) * (n ? n : 1)));
    if (n) {
        hipError_t e = hipMemcpy(fresh, src, sizeof(T) * n, hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(fresh); pt_set_error("hipMemcpy failed: %s", hipGetErrorString(e)); return PT_ERROR_HIP; }
    }
    (void)hipFree(*dst);
    *dst = fresh; *n_dst = n;
    return PT_OK;
}

static void lane_free(PtLane& l)
{
    for (int a = 0; a < 2; ++a) for (int k = 0; k < 4; ++k) (void)hipFree(l.q[a][k]);
    (void)hipFree(l.hits); (void)hipFree(l.path_result); (void)hipFree(l.counters); (void)hipFree(l.spill);
    if (l.done) (void)hipEventDestroy(l.done);
    if (l.accumulated) (void)hipEventDestroy(l.accumulated);
    if (l.stream) (void)hipStreamDestroy(l.stream);
    l = PtLane();
}

int pt_lanes_sync(PtContext* c)
{
    PT_HIP_CHECK(hipSetDevice(c->device));
    for (PtLane& l : c->lanes) if (l.stream) PT_HIP_CHECK(hipStreamSynchronize(l.stream));
    for (PtLane& l : c->lanes) l.pending = false;
    c->last_accumulate = nullptr;
    PT_HIP_CHECK(hipStreamSynchronize(c->stream));
    return PT_OK;
}

int pt_lanes_join(PtContext* c)
{
    for (PtLane& l : c->lanes) {
        if (!l.pending) continue;
        PT_HIP_CHECK(hipEventRecord(l.done, l.stream));
        PT_HIP_CHECK(hipStreamWaitEvent(c->stream, l.done, 0));
        l.pending = false;
    }
    return PT_OK;
}

int pt_lanes_resize(PtContext* c, int n)
{
    int rc = pt_lanes_sync(c);
    if (rc) return rc;
    for (PtLane& l : c->lanes) lane_free(l);
    c->lanes.assign((size_t)n, PtLane());
    c->next_lane = 0; c->last_accumulate = nullptr;
    for (PtLane& l : c->lanes) {
        PT_HIP_CHECK(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
        PT_HIP_CHECK(hipEventCreateWithFlags(&l.done, hipEventDisableTiming));
        PT_HIP_CHECK(hipEventCreateWithFlags(&l.accumulated, hipEventDisableTiming));
    }
    return PT_OK;
}

int pt_lane_begin(PtContext* c, PtLane** out)
{
    PtLane& l = c->lanes[c->next_lane % c->lanes.size()];
    c->next_lane++;
    // whatever the caller enqueued on its stream so far (uploads, a resolve that reads the image) comes first
    PT_HIP_CHECK(hipEventRecord(c->start_ev, c->stream));
    PT_HIP_CHECK(hipStreamWaitEvent(l.stream, c->start_ev, 0));
    l.pending = true;
    *out = &l;
    return PT_OK;
}

extern "C" {

const char* pt_last_error(void) { return g_error; }

int pt_ctx_create(int device_id, void* hip_stream, PtContext** out)
{
    PT_REQUIRE(out, "null argument");
    int count = 0;
    PT_HIP_CHECK(hipGetDeviceCount(&count));
    PT_REQUIRE(device_id >= 0 && device_id < count, "no such device");
    PT_HIP_CHECK(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    PT_HIP_CHECK(hipGetDeviceProperties(&prop, device_id));
    PtContext* c = new PtContext();
    c->device = device_id;
    c->num_cus = prop.multiProcessorCount;
    if (const char* e = getenv("PT_HIP_TRAVERSAL")) {
        c->traversal_mode = strcmp(e, "plain") == 0 ? 0 : (strcmp(e, "sched") == 0 ? 1 : 2);
#ifdef PT_EXPERIMENTAL
        if (strcmp(e, "pair") == 0) c->traversal_mode = 3;
        if (strcmp(e, "pool") == 0) c->traversal_mode = 4;
        if (strcmp(e, "autopool") == 0) c->traversal_mode = 5;
#endif
    }
    if (const char* e = getenv("PT_HIP_SCHED_MIN_RAYS")) c->sched_min_rays = (uint32_t)strtoul(e, nullptr, 10);
#ifdef PT_EXPERIMENTAL
    if (const char* e = getenv("PT_HIP_POOL_MIN_RAYS")) c->pool_min_rays = (uint32_t)strtoul(e, nullptr, 10);
#endif
    if (hip_stream) { c->stream = (hipStream_t)hip_stream; c->own_stream = false; }
    else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; pt_set_error("hipStreamCreate failed: %s", hipGetErrorString(e)); return PT_ERROR_HIP; }
        c->own_stream = true;
    }
    int lanes = 3;
    if (const char* e = getenv("PT_HIP_FRAMES_IN_FLIGHT")) { lanes = atoi(e); if (lanes < 1) lanes = 1; if (lanes > 8) lanes = 8; }
    hipError_t ee = hipEventCreateWithFlags(&c->start_ev, hipEventDisableTiming);
    if (ee != hipSuccess || pt_lanes_resize(c, lanes) != PT_OK) {
        if (ee != hipSuccess) pt_set_error("hipEventCreate failed: %s", hipGetErrorString(ee));
        pt_ctx_destroy(c);
        return PT_ERROR_HIP;
    }
    if (int rc = pt_noalloc_setup(c)) { pt_ctx_destroy(c); return rc; }
    *out = c;
    return PT_OK;
}

int pt_ctx_set_frames_in_flight(PtContext* c, int n)
{
    PT_REQUIRE(c, "null argument");
    PT_REQUIRE(n >= 1 && n <= 8, "frames in flight must be in [1, 8]");
    return pt_lanes_resize(c, n);
}

int pt_ctx_destroy(PtContext* c)
{
    if (!c) return PT_OK;
    (void)hipSetDevice(c->device);
    (void)pt_lanes_sync(c);
    for (PtLane& l : c->lanes) lane_free(l);
    if (c->start_ev) (void)hipEventDestroy(c->start_ev);
    PtWorkspace& w = c->ws;
    (void)hipFree(w.frame_result); (void)hipFree(w.stats);
    (void)hipFree(c->materials); (void)hipFree(c->geometry_nodes); (void)hipFree(c->blas_instances); (void)hipFree(c->textures);
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    for (auto& ch : c->build_arena.chunks) (void)hipFree(ch.first);
    if (c->pinned_word) (void)hipHostFree(c->pinned_word);
    if (c->mailbox) (void)hipHostFree(c->mailbox);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return PT_OK;
}

int pt_sync(PtContext* c)
{
    PT_REQUIRE(c, "null argument");
    return pt_lanes_sync(c);
}

int pt_trace_fence(PtContext* c)
{
    PT_REQUIRE(c, "null argument");
    PT_HIP_CHECK(hipSetDevice(c->device));
    return pt_lanes_join(c);
}

int pt_ctx_set_max_paths(PtContext* c, uint64_t max_paths)
{
    PT_REQUIRE(c, "null argument");
    PT_REQUIRE(max_paths >= 64 && max_paths < (1ull << 31), "max_paths must be in [64, 2^31)");
    c->max_paths = max_paths;
    return PT_OK;
}

int pt_buffer_upload(PtContext* c, const void* host, size_t bytes, void** dev_out)
{
    PT_REQUIRE(c && dev_out && (host || bytes == 0), "null argument");
    PT_HIP_CHECK(hipSetDevice(c->device));
    void* d = nullptr;
    PT_HIP_CHECK(hipMalloc(&d, bytes ? bytes : 16));
    if (bytes) {
        hipError_t e = hipMemcpy(d, host, bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(d); pt_set_error("hipMemcpy failed: %s", hipGetErrorString(e)); return PT_ERROR_HIP; }
    }
    *dev_out = d;
    return PT_OK;
}

int pt_buffer_download(PtContext* c, const void* dev, size_t bytes, void* host_out)
{
    PT_REQUIRE(c && dev && host_out, "null argument");
    int rc = pt_lanes_sync(c);
    if (rc) return rc;
    PT_HIP_CHECK(hipMemcpy(host_out, dev, bytes, hipMemcpyDeviceToHost));
    return PT_OK;
}

int pt_buffer_free(PtContext* c, void* dev)
{
    PT_REQUIRE(c, "null argument");
    int rc = pt_lanes_sync(c);   // a frame in flight may still read or write the buffer
    if (rc) return rc;
    PT_HIP_CHECK(hipFree(dev));
    return PT_OK;
}

int pt_scene_set_tables(PtContext* c,
                        const PtMaterial* materials, uint32_t n_materials,
                        const PtGeometryNode* geometry_nodes, uint32_t n_geometry_nodes,
                        const PtBLASInstance* blas_instances, uint32_t n_blas_instances,
                        const PtTextureDesc* textures, uint32_t n_textures)
{
    PT_REQUIRE(c, "null argument");
    PT_REQUIRE((materials || !n_materials) && (geometry_nodes || !n_geometry_nodes) && (blas_instances || !n_blas_instances) && (textures || !n_textures), "null table");
    // the reference logs and skips the upload above 1024 entries (source/resources/bindless_resources.cpp:85-89)
    if (n_materials > 1024 || n_geometry_nodes > 1024 || n_blas_instances > 1024 || n_textures > 1024) {
        pt_set_error("pt_scene_set_tables: a table exceeds 1024 entries");
        return PT_ERROR_LIMIT;
    }
    // every index a shader will follow is checked here once instead of per hit
    for (uint32_t i = 0; i < n_blas_instances; ++i)
        PT_REQUIRE(blas_instances[i].firstGeometryIndex < n_geometry_nodes, "BLASInstance.firstGeometryIndex out of range");
    for (uint32_t i = 0; i < n_geometry_nodes; ++i) {
        PT_REQUIRE(geometry_nodes[i].materialIndex < n_materials, "GeometryNode.materialIndex out of range");
        PT_REQUIRE(geometry_nodes[i].vertexBufferDeviceAddress && geometry_nodes[i].indexBufferDeviceAddress, "GeometryNode with a null buffer address");
    }
    for (uint32_t i = 0; i < n_materials; ++i)
        if (materials[i].useAlbedoMap) PT_REQUIRE(materials[i].albedoMapIndex < n_textures, "Material.albedoMapIndex out of range");
    for (uint32_t i = 0; i < n_textures; ++i)
        PT_REQUIRE(textures[i].rgba8_dev && textures[i].width && textures[i].height, "empty texture");
    int rc = pt_lanes_sync(c);
    if (rc) return rc;
    if ((rc = replace_table(&c->materials, &c->n_materials, materials, n_materials))) return rc;
    if ((rc = replace_table(&c->geometry_nodes, &c->n_geometry_nodes, geometry_nodes, n_geometry_nodes))) return rc;
    if ((rc = replace_table(&c->blas_instances, &c->n_blas_instances, blas_instances, n_blas_instances))) return rc;
    if ((rc = replace_table(&c->textures, &c->n_textures, textures, n_textures))) return rc;
    c->h_materials.assign(materials, materials + n_materials);
    c->h_geometry_nodes.assign(geometry_nodes, geometry_nodes + n_geometry_nodes);
    c->h_blas_instances.assign(blas_instances, blas_instances + n_blas_instances);
    c->h_textures.assign(textures, textures + n_textures);
    c->tables_version++;
    return PT_OK;
}

} // extern "C"
