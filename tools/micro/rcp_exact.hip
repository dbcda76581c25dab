// rcp_exact.hip -- over every binary32 bit pattern: do the short forms of 1/x and sqrt(x) built on the hardware
// approximations (v_rcp_f32, v_sqrt_f32, v_rsq_f32) and two or three fmas give the bits of the correctly rounded
// operations (the compiler's IEEE expansions, which a CPU's division and square root agree with)?
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o rcp_exact rcp_exact.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int KIND> __device__ __forceinline__ float candidate(float x)
{
    if (KIND == 0) {            // rcp + two fmas
        const float r = __builtin_amdgcn_rcpf(x);
        const float e = __builtin_fmaf(-x, r, 1.0f);
        return __builtin_fmaf(e, r, r);
    }
    if (KIND == 1) {            // rcp + three fmas: the error term carries its square
        const float r = __builtin_amdgcn_rcpf(x);
        float e = __builtin_fmaf(-x, r, 1.0f);
        e = __builtin_fmaf(e, e, e);
        return __builtin_fmaf(e, r, r);
    }
    if (KIND == 2) {            // rcp + two Newton steps
        float r = __builtin_amdgcn_rcpf(x);
        float e = __builtin_fmaf(-x, r, 1.0f);
        r = __builtin_fmaf(e, r, r);
        e = __builtin_fmaf(-x, r, 1.0f);
        return __builtin_fmaf(e, r, r);
    }
    if (KIND == 3) {            // sqrt: s = x * rsq(x), then one correction with the half reciprocal root
        const float y = __builtin_amdgcn_rsqf(x);
        const float s = x * y, h = 0.5f * y;
        const float d = __builtin_fmaf(-s, s, x);
        return __builtin_fmaf(d, h, s);
    }
    if (KIND == 4) {            // sqrt: hardware root, one correction
        const float s = __builtin_amdgcn_sqrtf(x);
        const float h = 0.5f * __builtin_amdgcn_rcpf(s);
        const float d = __builtin_fmaf(-s, s, x);
        return __builtin_fmaf(d, h, s);
    }
    if (KIND == 5 || KIND == 6) {   // x / c for a constant c: product with the rounded reciprocal, one correction
        const float c = KIND == 5 ? 3.14159265f : 255.0f, rc = 1.0f / c;
        const float q = x * rc;
        const float rem = __builtin_fmaf(-c, q, x);
        return __builtin_fmaf(rem, rc, q);
    }
    return 0.0f;
}
template <int KIND> __device__ __forceinline__ float reference(float x)
{
    if (KIND == 5) return x / 3.14159265f;
    if (KIND == 6) return x / 255.0f;
    return KIND >= 3 ? sqrtf(x) : 1.0f / x;
}

// stats[0] mismatches inside [lo, hi] (magnitudes), [1] first mismatching pattern + 1, [2..3] smallest and largest magnitude bits that mismatch
template <int KIND>
__global__ void k_check(uint32_t lo_bits, uint32_t hi_bits, unsigned long long* stats)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const uint32_t bits = (uint32_t)i, mag = bits & 0x7FFFFFFFu;
        if (mag < lo_bits || mag > hi_bits) continue;
        if ((KIND == 3 || KIND == 4) && (bits >> 31)) continue;
        const float x = __uint_as_float(bits);
        const float a = candidate<KIND>(x), b = reference<KIND>(x);
        if (__float_as_uint(a) != __float_as_uint(b) && !(a != a && b != b)) {
            atomicAdd(&stats[0], 1ull);
            atomicMin(&stats[2], (unsigned long long)mag);
            atomicMax(&stats[3], (unsigned long long)mag);
            atomicCAS(&stats[1], 0ull, (unsigned long long)bits + 1ull);
        }
    }
}

template <int KIND> int run(const char* name, float lo, float hi)
{
    unsigned long long* d; CHECK(hipMalloc((void**)&d, 32));
    unsigned long long h[4] = { 0, 0, ~0ull, 0 };
    CHECK(hipMemcpy(d, h, 32, hipMemcpyHostToDevice));
    uint32_t lo_bits, hi_bits; memcpy(&lo_bits, &lo, 4); memcpy(&hi_bits, &hi, 4);
    k_check<KIND><<<4096, 256>>>(lo_bits, hi_bits, d);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h, d, 32, hipMemcpyDeviceToHost));
    printf("%-34s |x| in [%g, %g]: %llu mismatches", name, lo, hi, h[0]);
    if (h[0]) printf(", first pattern 0x%08llx, magnitudes 0x%08llx .. 0x%08llx", h[1] - 1, h[2], h[3]);
    printf("\n");
    CHECK(hipFree(d));
    return 0;
}

int main()
{
    const float lo[] = { 0x1p-100f, 0x1p-126f, 0.0f, 0.0f }, hi[] = { 0x1p100f, 0x1p126f, __builtin_inff(), 0.0f };
    for (int k = 0; k < 4; ++k) {
        if (run<0>("rcp + 2 fma", lo[k], hi[k])) return 1;
        if (run<1>("rcp + 3 fma (e + e^2)", lo[k], hi[k])) return 1;
        if (run<2>("rcp + two Newton steps (4 fma)", lo[k], hi[k])) return 1;
        if (run<3>("sqrt by rsq, one correction", lo[k], hi[k])) return 1;
        if (run<4>("sqrt by sqrt + rcp, one correction", lo[k], hi[k])) return 1;
        if (run<5>("x / pi by reciprocal, one correction", lo[k], hi[k])) return 1;
        if (run<6>("x / 255 by reciprocal, one correction", lo[k], hi[k])) return 1;
    }
    return 0;
}
