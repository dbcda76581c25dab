/* or_integrator.cpp -- ray_gen / closest_hit / miss restated on the CPU
 * (test infrastructure only; see pt_oracle.h for the cited reference lines). */
#include "or_internal.h"
#include <thread>
#include <atomic>
#include <limits>

extern "C" {

OrScene* or_scene_create(const OrTlas* tlas,
                         const OrMaterial* materials, uint32_t n_mat,
                         const OrGeometryNode* geometry_nodes, uint32_t n_geo,
                         const OrBLASInstance* blas_instances, uint32_t n_inst,
                         const OrTexture* textures, uint32_t n_tex)
{
    OrScene* s = new OrScene();
    s->tlas = tlas;
    s->materials.assign(materials, materials + n_mat);
    s->geometry_nodes.assign(geometry_nodes, geometry_nodes + n_geo);
    s->blas_instances.assign(blas_instances, blas_instances + n_inst);
    s->textures.assign(textures, textures + n_tex);
    return s;
}
void or_scene_destroy(OrScene* s) { delete s; }

/* texture(textures[i], uv) of closest_hit.rchit:78 with the reference's sampler state
 * (include/resources/gpu_resources.hpp:44-63: LINEAR min/mag, REPEAT, one mip level) on an
 * R8G8B8A8_UNORM image (source/model_loader.cpp:18): texel centres at +0.5, full-precision
 * weights, lerp(a,b,w) = fma(w, b - a, a) along x then y. */
float or_texture_bilinear_channel(const OrTexture* tex, float u, float v, int ch)
{
    const int W = (int)tex->width, H = (int)tex->height;
    float U = or_fma(u, (float)W, -0.5f), Vv = or_fma(v, (float)H, -0.5f);
    float fu = floorf(U), fv = floorf(Vv);
    float a = U - fu, b = Vv - fv;
    long i0 = (long)fu, j0 = (long)fv;
    auto wrap = [](long i, int n) { long m = i % n; return (int)(m < 0 ? m + n : m); };
    int x0 = wrap(i0, W), x1 = wrap(i0 + 1, W), y0 = wrap(j0, H), y1 = wrap(j0 + 1, H);
    auto tx = [&](int x, int y) { return (float)tex->rgba8[((size_t)y * W + x) * 4 + ch] / 255.0f; };
    float c00 = tx(x0, y0), c10 = tx(x1, y0), c01 = tx(x0, y1), c11 = tx(x1, y1);
    float r0 = or_fma(a, c10 - c00, c00);
    float r1 = or_fma(a, c11 - c01, c01);
    return or_fma(b, r1 - r0, r0);
}

} /* extern "C" */

struct Payload {            /* shaders/ray.glsl:1-9 */
    V3 hitValue; uint32_t seed; uint32_t depth; V3 rayOrigin; V3 rayDirection; V3 weight;
};

static const float PI_F = 3.14159265f;

/* shaders/closest_hit.rchit:55-86 */
static void closest_hit_shader(const OrScene* sc, const HitRec& h, Payload& p, OrStats* st)
{
    const OrInst& I = sc->tlas->inst[h.inst];
    const OrBLASInstance& bi = sc->blas_instances[I.custom_index];
    const OrGeometryNode& gn = sc->geometry_nodes[bi.firstGeometryIndex + 0];
    const OrMaterial& m = sc->materials[gn.materialIndex];
    const OrVertex* vertices = (const OrVertex*)(uintptr_t)gn.vertexBufferDeviceAddress;
    const uint32_t* indices = (const uint32_t*)(uintptr_t)gn.indexBufferDeviceAddress;
    const OrVertex& a = vertices[indices[h.prim * 3 + 0]];
    const OrVertex& b = vertices[indices[h.prim * 3 + 1]];
    const OrVertex& c = vertices[indices[h.prim * 3 + 2]];
    const float b0 = 1.0f - h.u - h.v, b1 = h.u, b2 = h.v;
    auto mix3 = [&](float x, float y, float z) { return or_fma(z, b2, or_fma(y, b1, x * b0)); };
    V3 pos { mix3(a.pos[0], b.pos[0], c.pos[0]), mix3(a.pos[1], b.pos[1], c.pos[1]), mix3(a.pos[2], b.pos[2], c.pos[2]) };
    V3 nrm = or_normalize(V3 { mix3(a.normal[0], b.normal[0], c.normal[0]), mix3(a.normal[1], b.normal[1], c.normal[1]), mix3(a.normal[2], b.normal[2], c.normal[2]) });
    float tu = mix3(a.uv[0], b.uv[0], c.uv[0]), tv = mix3(a.uv[1], b.uv[1], c.uv[1]);

    V3 worldPos = or_xform_point(I.o2w, pos);
    /* vec3(normal * gl_WorldToObjectEXT): component j = dot(normal, column j) */
    const float* w = I.w2o;
    V3 wn = or_normalize(V3 {
        or_fma(w[0], nrm.x, or_fma(w[4], nrm.y, w[8] * nrm.z)),
        or_fma(w[1], nrm.x, or_fma(w[5], nrm.y, w[9] * nrm.z)),
        or_fma(w[2], nrm.x, or_fma(w[6], nrm.y, w[10] * nrm.z)) });

    float N[3] = { wn.x, wn.y, wn.z }, T[3], B[3], dir[3];
    or_create_coordinate_system(N, T, B);
    or_sampling_hemisphere(&p.seed, T, B, N, dir);
    float cosTheta = or_fma(dir[2], N[2], or_fma(dir[1], N[1], dir[0] * N[0]));
    float pdf = cosTheta / PI_F;

    float albedo[3] = { m.albedoFactor[0], m.albedoFactor[1], m.albedoFactor[2] };
    if (m.useAlbedoMap) {
        const OrTexture* tex = &sc->textures[m.albedoMapIndex];
        for (int k = 0; k < 3; ++k) albedo[k] = albedo[k] * or_texture_bilinear_channel(tex, tu, tv, k);
        if (st) st->tex_hits++;
    }
    p.rayOrigin = worldPos;
    p.rayDirection = V3 { dir[0], dir[1], dir[2] };
    p.hitValue = V3 { m.emissiveFactor[0], m.emissiveFactor[1], m.emissiveFactor[2] };
    p.weight = V3 { (albedo[0] / PI_F) * cosTheta / pdf, (albedo[1] / PI_F) * cosTheta / pdf, (albedo[2] / PI_F) * cosTheta / pdf };
}

/* shaders/miss.rmiss:8-21 */
static void miss_shader(Payload& p)
{
    if (p.depth == 0) p.hitValue = V3 { 1.0f * 0.8f, 1.0f * 0.8f, 1.0f * 0.8f };
    else p.hitValue = V3 { 0.01f, 0.01f, 0.01f };
    p.depth = 100;
}

/* one iteration of the sample loop, shaders/ray_gen.rgen:40-71 (returns hitValue before * hitStrength) */
static V3 trace_one_path(const OrScene* sc, V3 origin, V3 direction, uint32_t seed, uint32_t max_depth,
                         bool use_bvh, OrStats* st, OrBounce* rec, uint32_t* n_rec)
{
    Payload p;
    p.hitValue = V3 { 0, 0, 0 }; p.seed = seed; p.depth = 0;
    p.rayOrigin = origin; p.rayDirection = direction; p.weight = V3 { 0, 0, 0 };
    V3 hitValue { 0, 0, 0 }, cw { 1, 1, 1 };
    uint32_t nb = 0;
    for (; p.depth < max_depth; ++p.depth) {
        HitRec h;
        V3 ro = p.rayOrigin, rd = p.rayDirection;
        or_closest_hit(sc->tlas, ro, rd, 0.001f, 10000.0f, use_bvh, h, st);
        if (h.hit) closest_hit_shader(sc, h, p, st);
        else miss_shader(p);
        if (rec) {
            OrBounce& r = rec[nb];
            r.hit.instance = h.hit ? h.inst : 0xFFFFFFFFu; r.hit.primitive = h.hit ? h.prim : 0xFFFFFFFFu;
            r.hit.u = h.u; r.hit.v = h.v; r.t = h.t;
            r.origin[0] = ro.x; r.origin[1] = ro.y; r.origin[2] = ro.z;
            r.dir[0] = rd.x; r.dir[1] = rd.y; r.dir[2] = rd.z;
            r.hit_value[0] = p.hitValue.x; r.hit_value[1] = p.hitValue.y; r.hit_value[2] = p.hitValue.z;
            r.weight[0] = p.weight.x; r.weight[1] = p.weight.y; r.weight[2] = p.weight.z;
            r.seed_after = p.seed;
        }
        ++nb;
        hitValue = V3 { or_fma(p.hitValue.x, cw.x, hitValue.x), or_fma(p.hitValue.y, cw.y, hitValue.y), or_fma(p.hitValue.z, cw.z, hitValue.z) };
        cw = V3 { cw.x * p.weight.x, cw.y * p.weight.y, cw.z * p.weight.z };
    }
    if (n_rec) *n_rec = nb;
    if (st) st->paths++;
    return hitValue;
}

static void render_rows(const OrScene* sc, const float* cam, uint32_t W, uint32_t H, uint32_t frame, uint32_t spp,
                        uint32_t max_depth, uint32_t x0, uint32_t x1, std::atomic<uint32_t>* next_row, uint32_t y1,
                        bool use_bvh, float* result, OrStats* st)
{
    for (;;) {
        uint32_t y = next_row->fetch_add(1);
        if (y >= y1) break;
        for (uint32_t x = x0; x < x1; ++x) {
            float o[3], d[3];
            or_primary_ray(cam, W, H, x, y, o, d);
            V3 res { 0, 0, 0 };
            for (uint32_t i = 0; i < spp; ++i) {
                uint32_t seed = or_tea(y * W + x, frame * spp + i);
                V3 hv = trace_one_path(sc, V3 { o[0], o[1], o[2] }, V3 { d[0], d[1], d[2] }, seed, max_depth, use_bvh, st, nullptr, nullptr);
                res = V3 { or_fma(hv.x, 2.5f, res.x), or_fma(hv.y, 2.5f, res.y), or_fma(hv.z, 2.5f, res.z) };
            }
            float fs = (float)spp;
            float* out = result + ((size_t)y * W + x) * 3;
            out[0] = res.x / fs; out[1] = res.y / fs; out[2] = res.z / fs;
        }
    }
}

extern "C" {

void or_render_frame(const OrScene* scene, const float camera[32], uint32_t W, uint32_t H,
                     uint32_t frame, uint32_t spp, uint32_t max_depth,
                     uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                     int use_bvh, int threads, float* result, OrStats* stats)
{
    if (threads < 1) threads = 1;
    std::atomic<uint32_t> next_row(y0);
    std::vector<OrStats> st(threads);
    memset(st.data(), 0, sizeof(OrStats) * threads);
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t)
        pool.emplace_back(render_rows, scene, camera, W, H, frame, spp, max_depth, x0, x1, &next_row, y1, use_bvh != 0, result, &st[t]);
    render_rows(scene, camera, W, H, frame, spp, max_depth, x0, x1, &next_row, y1, use_bvh != 0, result, &st[0]);
    for (auto& th : pool) th.join();
    if (stats) {
        for (int t = 0; t < threads; ++t) {
            stats->rays += st[t].rays; stats->box_tests += st[t].box_tests; stats->tri_tests += st[t].tri_tests;
            stats->hits += st[t].hits; stats->tex_hits += st[t].tex_hits; stats->paths += st[t].paths;
        }
    }
}

uint32_t or_trace_path(const OrScene* scene, const float camera[32], uint32_t W, uint32_t H,
                       uint32_t px, uint32_t py, uint32_t sample_index, uint32_t max_depth,
                       OrBounce* bounces, float radiance[3])
{
    float o[3], d[3];
    or_primary_ray(camera, W, H, px, py, o, d);
    uint32_t seed = or_tea(py * W + px, sample_index);
    uint32_t nb = 0;
    V3 hv = trace_one_path(scene, V3 { o[0], o[1], o[2] }, V3 { d[0], d[1], d[2] }, seed, max_depth, true, nullptr, bounces, &nb);
    radiance[0] = hv.x; radiance[1] = hv.y; radiance[2] = hv.z;
    return nb;
}

/* shaders/ray_gen.rgen:76-87 */
void or_accumulate(int mode, uint32_t n_pixels, uint32_t frame, const float* result, void* accum)
{
    if (mode == 0) {
        float* a = (float*)accum;
        for (uint32_t i = 0; i < n_pixels; ++i) {
            a[4 * i + 0] += result[3 * i + 0]; a[4 * i + 1] += result[3 * i + 1]; a[4 * i + 2] += result[3 * i + 2];
            a[4 * i + 3] += 1.0f;
        }
    } else if (mode == 1) {
        float* a = (float*)accum;
        float w = 1.0f / (float)(frame + 1);
        for (uint32_t i = 0; i < n_pixels; ++i) {
            for (int k = 0; k < 3; ++k) {
                float old = a[4 * i + k], r = result[3 * i + k];
                a[4 * i + k] = frame > 0 ? or_fma(r, w, old * (1.0f - w)) : r;
            }
            a[4 * i + 3] = 1.0f;
        }
    } else {
        uint8_t* a = (uint8_t*)accum;
        float w = 1.0f / (float)(frame + 1);
        for (uint32_t i = 0; i < n_pixels; ++i) {
            for (int k = 0; k < 3; ++k) {
                float old = (float)a[4 * i + k] / 255.0f, r = result[3 * i + k];
                float c = frame > 0 ? or_fma(r, w, old * (1.0f - w)) : r;
                c = c > 0.0f ? (c < 1.0f ? c : 1.0f) : 0.0f;   /* NaN -> 0 */
                a[4 * i + k] = (uint8_t)(int)rintf(c * 255.0f);
            }
            a[4 * i + 3] = 255;
        }
    }
}

void or_resolve(uint32_t n_pixels, const float* s, uint8_t* rgba8)
{
    for (uint32_t i = 0; i < n_pixels; ++i) {
        float n = s[4 * i + 3];
        for (int k = 0; k < 3; ++k) {
            float c = s[4 * i + k] / n;
            c = c > 0.0f ? (c < 1.0f ? c : 1.0f) : 0.0f;
            rgba8[4 * i + k] = (uint8_t)(int)rintf(c * 255.0f);
        }
        rgba8[4 * i + 3] = 255;
    }
}

} /* extern "C" */
