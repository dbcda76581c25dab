/* or_sampling.cpp -- CPU restatement of shaders/sampling.glsl (test infrastructure only). */
#include "or_internal.h"

extern "C" {

/* shaders/sampling.glsl:4-18 */
uint32_t or_tea(uint32_t val0, uint32_t val1)
{
    uint32_t v0 = val0, v1 = val1, s0 = 0;
    for (uint32_t n = 0; n < 16; ++n) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}

/* shaders/sampling.glsl:22-28 */
uint32_t or_lcg(uint32_t* prev)
{
    *prev = 1664525u * (*prev) + 1013904223u;
    return *prev & 0x00FFFFFFu;
}

/* shaders/sampling.glsl:31-34; the 24-bit integer converts exactly and 2^-24 scales exactly */
float or_random(uint32_t* prev)
{
    return (float)or_lcg(prev) / (float)0x01000000;
}

/* Pinned sin/cos for x >= 0 (the shader's cos()/sin() are driver-defined, shaders/sampling.glsl:46).
 * Three-term Cody-Waite reduction by pi/2 followed by the classic single-precision
 * minimax polynomials on [-pi/4, pi/4]; every step is an explicit fma. */
void or_sincos(float x, float* s_out, float* c_out)
{
    int k = (int)(x * 0.636619772f + 0.5f);
    float kf = (float)k;
    float r = or_fma(kf, -1.5703125f, x);
    r = or_fma(kf, -4.837512969970703125e-4f, r);
    r = or_fma(kf, -7.54978995489188216e-8f, r);
    float r2 = r * r;
    float ps = or_fma(or_fma(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f);
    float s = or_fma(r * r2, ps, r);
    float pc = or_fma(or_fma(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f);
    float c = or_fma(r2 * r2, pc, or_fma(-0.5f, r2, 1.0f));
    switch (k & 3) {
    case 0: *s_out = s;  *c_out = c;  break;
    case 1: *s_out = c;  *c_out = -s; break;
    case 2: *s_out = -s; *c_out = -c; break;
    default: *s_out = -c; *c_out = s; break;
    }
}

/* shaders/sampling.glsl:53-65; "vec3 / float" is evaluated as a multiply by the reciprocal */
void or_create_coordinate_system(const float N[3], float Nt[3], float Nb[3])
{
    if (fabsf(N[0]) > fabsf(N[1])) {
        float inv = 1.0f / sqrtf(or_fma(N[2], N[2], N[0] * N[0]));
        Nt[0] = N[2] * inv; Nt[1] = 0.0f; Nt[2] = -N[0] * inv;
    } else {
        float inv = 1.0f / sqrtf(or_fma(N[2], N[2], N[1] * N[1]));
        Nt[0] = 0.0f; Nt[1] = -N[2] * inv; Nt[2] = N[1] * inv;
    }
    /* cross(N, Nt) */
    Nb[0] = or_fma(N[1], Nt[2], -(N[2] * Nt[1]));
    Nb[1] = or_fma(N[2], Nt[0], -(N[0] * Nt[2]));
    Nb[2] = or_fma(N[0], Nt[1], -(N[1] * Nt[0]));
}

/* shaders/sampling.glsl:38-50; r1 is drawn before r2; 2*PI with PI = 3.14159265 rounds to 6.28318548f */
void or_sampling_hemisphere(uint32_t* seed, const float x[3], const float y[3], const float z[3], float out[3])
{
    float r1 = or_random(seed);
    float r2 = or_random(seed);
    float sq = sqrtf(r1);
    float s, c;
    or_sincos(6.28318548f * r2, &s, &c);
    float dx = c * sq, dy = s * sq, dz = sqrtf(1.0f - r1);
    for (int i = 0; i < 3; ++i)
        out[i] = or_fma(dz, z[i], or_fma(dy, y[i], dx * x[i]));
}

/* shaders/ray_gen.rgen:21-27 */
void or_primary_ray(const float cam[32], uint32_t W, uint32_t H, uint32_t px, uint32_t py, float o[3], float d[3])
{
    const float* VI = cam;       /* column-major: VI[c*4 + r] */
    const float* PI = cam + 16;
    float uvx = ((float)px + 0.5f) / (float)W;
    float uvy = ((float)py + 0.5f) / (float)H;
    float dx = uvx * 2.0f - 1.0f;
    float dy = uvy * 2.0f - 1.0f;
    o[0] = VI[12]; o[1] = VI[13]; o[2] = VI[14];
    V3 t;
    t.x = or_fma(PI[0], dx, or_fma(PI[4], dy, PI[8] + PI[12]));
    t.y = or_fma(PI[1], dx, or_fma(PI[5], dy, PI[9] + PI[13]));
    t.z = or_fma(PI[2], dx, or_fma(PI[6], dy, PI[10] + PI[14]));
    V3 n = or_normalize(t);
    d[0] = or_fma(VI[0], n.x, or_fma(VI[4], n.y, VI[8] * n.z));
    d[1] = or_fma(VI[1], n.x, or_fma(VI[5], n.y, VI[9] * n.z));
    d[2] = or_fma(VI[2], n.x, or_fma(VI[6], n.y, VI[10] * n.z));
}

/* Inverse of a row-major 3x4 affine transform, evaluated in binary64 by cofactors and
 * rounded once to binary32 (the driver's gl_WorldToObjectEXT, closest_hit.rchit:63). */
void or_invert_transform(const float m[12], float w[12])
{
    double a00 = m[0], a01 = m[1], a02 = m[2], t0 = m[3];
    double a10 = m[4], a11 = m[5], a12 = m[6], t1 = m[7];
    double a20 = m[8], a21 = m[9], a22 = m[10], t2 = m[11];
    double c00 = a11 * a22 - a12 * a21;
    double c01 = a12 * a20 - a10 * a22;
    double c02 = a10 * a21 - a11 * a20;
    double det = (a00 * c00 + a01 * c01) + a02 * c02;
    double i00 = c00 / det, i01 = (a02 * a21 - a01 * a22) / det, i02 = (a01 * a12 - a02 * a11) / det;
    double i10 = c01 / det, i11 = (a00 * a22 - a02 * a20) / det, i12 = (a02 * a10 - a00 * a12) / det;
    double i20 = c02 / det, i21 = (a01 * a20 - a00 * a21) / det, i22 = (a00 * a11 - a01 * a10) / det;
    w[0] = (float)i00; w[1] = (float)i01; w[2] = (float)i02;  w[3]  = (float)(-((i00 * t0 + i01 * t1) + i02 * t2));
    w[4] = (float)i10; w[5] = (float)i11; w[6] = (float)i12;  w[7]  = (float)(-((i10 * t0 + i11 * t1) + i12 * t2));
    w[8] = (float)i20; w[9] = (float)i21; w[10] = (float)i22; w[11] = (float)(-((i20 * t0 + i21 * t1) + i22 * t2));
}

} /* extern "C" */
