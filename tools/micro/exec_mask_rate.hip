// exec_mask_rate.hip -- does a wave64 vector instruction cost less when whole quarters (16 lanes) or halves of the
// wave are masked off?  Chains of v_fma_f32 (fast class) and v_max_f32 (slow class) run with the lanes below N active,
// N = 16, 32, 48, 64, and with every fourth lane active (16 lanes spread over the wave), 4 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o exec_mask_rate exec_mask_rate.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int kIters = 4096;

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, float a, float b, int n_active, int spread)
{
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = (float)threadIdx.x + (float)i;
    const int lane = threadIdx.x & 63;
    const bool on = spread ? (lane % spread) == 0 : lane < n_active;
    if (on) {
        for (int it = 0; it < kIters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                if (KIND == 1) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
            }
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND> int run(const char* name, int cus)
{
    float* out; CHECK(hipMalloc((void**)&out, sizeof(float) * 256 * cus * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int cases[][2] = { {64, 0}, {48, 0}, {32, 0}, {16, 0}, {8, 0}, {0, 4}, {0, 2} };
    for (auto& c : cases) {
        k<KIND><<<cus * 4, 256>>>(out, 1.0001f, 0.5f, c[0], c[1]);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        k<KIND><<<cus * 4, 256>>>(out, 1.0001f, 0.5f, c[0], c[1]);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (c[1]) printf("%-10s every %d. lane active: %.3f ms\n", name, c[1], ms);
        else printf("%-10s lanes 0..%d active:   %.3f ms\n", name, c[0] - 1, ms);
    }
    return 0;
}
int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    if (run<0>("v_fma_f32", p.multiProcessorCount)) return 1;
    if (run<1>("v_max_f32", p.multiProcessorCount)) return 1;
    return 0;
}
