// valu_rate.hip -- how many wave64 vector instructions per cycle one SIMD of gfx950 issues, for the
// instruction kinds the traversal kernels are made of, at 1, 2, 4 and 8 waves per SIMD.  Every lane
// runs 16 independent chains of one instruction kind; the rate "by event" is wave instructions issued
// on one SIMD divided by the kernel's duration in 2.4 GHz cycles, so it includes what the clock gives
// up under load.  Findings (MI355X): v_fma/mul/add/sub_f32, v_and/or/xor_b32, v_add/sub_u32, v_mov_b32
// and register shifts reach 0.35-0.42 per cycle from two waves up; v_cndmask_b32, every min/max/med3,
// v_cmp_*, v_cvt_f32_ubyteN, v_cvt_f32_u32, the three-operand integer forms (v_lshl_add_u32, v_add3_u32,
// v_and_or_b32, v_bfi_b32, v_perm_b32, v_bfe_u32), v_fmac_f32 and v_pk_fma_f32 stay at 0.22-0.24
// however many waves share the SIMD; v_rcp_f32 0.12.  A stream of nothing but v_cndmask_b32_e32 (implicit
// vcc) drops to 0.044, but mixed with other instructions the e32 and e64 forms cost the same.
// Sub-dword operand forms (`valu_rate s`): v_or/and/mov/lshlrev_b32_sdwa and v_add_f32_sdwa with a byte or word
// selected, v_mov_b32_dpp, v_cvt_f32_fp8, v_cvt_pk_f32_fp8 and v_cvt_scalef32_pk_f32_fp8 all issue at 0.22-0.23: a byte
// cannot be brought into a float register at the full rate by any of them.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kIters = 4096;

// 16 independent chains per lane, 16 instructions per iteration
template <int KIND>
__global__ void __launch_bounds__(256) k_rate(float* out, float a, float b, uint32_t w, unsigned long long* clocks)
{
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = (float)threadIdx.x + (float)i;
    uint32_t word = w + threadIdx.x;
    unsigned long long mask = w * 0x100000001ull + threadIdx.x / 64;
    double y[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) y[i] = (double)threadIdx.x + (double)i;
    if (KIND >= 40) asm volatile("s_mov_b64 vcc, 0x5555" ::: "vcc");
    const unsigned long long c0 = clock64();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
            if (KIND == 1) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(x[i]) : "v"(a) : "s20", "s21");
            if (KIND == 2) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(x[i]) : "v"(word));
            if (KIND == 3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
            if (KIND == 4) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x[i]), "v"(a) : "vcc");
            if (KIND == 5) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            if (KIND == 6) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
            if (KIND == 7) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
            if (KIND == 8) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[i]) : "v"(word));
            if (KIND == 9) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(word));
            if (KIND == 10) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(a));
            if (KIND == 11) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(x[i]) : "v"(word));
            if (KIND == 12) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(y[i & 7]) : "v"(y[8]));
            if (KIND == 40) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a));
            if (KIND == 41) asm volatile("v_cndmask_b32_e32 %0, %1, %0, vcc" : "+v"(x[i]) : "v"(a));
            if (KIND == 42) asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(x[i]) : "v"(a), "v"(b));
            if (KIND == 43) asm volatile("v_cndmask_b32_e64 %0, %1, %2, s[20:21]" : "=v"(x[i]) : "v"(a), "v"(b) : "s20", "s21");
            if (KIND == 44) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(x[i]) : "v"(word) : "vcc");
            if (KIND == 50) asm volatile("v_cndmask_b32_e32 %0, %1, %0, vcc\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            if (KIND == 51) asm volatile("v_cndmask_b32_e64 %0, %1, %0, vcc\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            if (KIND == 52) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %1, %0, vcc" : "+v"(x[i]) : "v"(a) : "vcc");
            if (KIND == 53) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32_e64 %0, %1, %0, vcc" : "+v"(x[i]) : "v"(a) : "vcc");
            if (KIND == 13) asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1" : : "v"(x[i]), "v"(a) : "s20", "s21");
            if (KIND == 14) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x[i]) : "v"(word));
            if (KIND == 15) asm volatile("v_max_i32 %0, %0, %1" : "+v"(x[i]) : "v"(word));
            if (KIND == 16) asm volatile("v_min3_u32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(word));
            if (KIND == 17) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(a));
            if (KIND == 18) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(x[i]) : "v"(word));
            if (KIND == 19) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(word));
            if (KIND == 20) asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(x[i]) : );
            if (KIND == 21) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(x[i]) : "v"(word));
            if (KIND == 22) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(word));
            if (KIND == 23) asm volatile("v_or_b32 %0, %0, %1" : "+v"(x[i]) : "v"(word));
            if (KIND == 24) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(word));
            if (KIND == 25) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            if (KIND == 26) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            if (KIND == 27) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x[i]) : );
            if (KIND == 28) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(x[i]) : );
            if (KIND == 29) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(x[i]), "v"(a) : "vcc");
            if (KIND == 30) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a) : );
            if (KIND == 31) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(x[i]) : "v"(a));
            if (KIND == 32) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(word));
            if (KIND == 33) asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(x[i]) : "v"(word));
            if (KIND == 34) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x[i]) : "v"(word));
            if (KIND == 35) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(word));
            if (KIND == 36) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]) : );
            if (KIND == 37) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(x[i]) : "v"(word));
            if (KIND == 38) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(x[i]) : );
            if (KIND == 39) asm volatile("v_cmp_class_f32 vcc, %0, %1" : : "v"(x[i]), "v"(a) : "vcc");
            if (KIND == 60) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(x[i]) : "v"(word), "v"(a));
            if (KIND == 61) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(x[i]) : "v"(word), "v"(a));
            if (KIND == 62) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(x[i]) : "v"(word));
            if (KIND == 63) asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(x[i]) : "v"(a));
            if (KIND == 64) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            if (KIND == 65) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            if (KIND == 66) asm volatile("v_cvt_f32_ubyte0_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(x[i]) : "v"(word));
            if (KIND == 67) asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n v_fma_f32 %0, %0, %2, %3" : "=v"(x[i]) : "v"(word), "v"(a), "v"(b));
            if (KIND == 68) asm volatile("v_cvt_f32_ubyte1 %0, %1\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %0, %0, %2, %3" : "=v"(x[i]) : "v"(word), "v"(a), "v"(b));
            if (KIND == 69) asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(x[i]) : "v"(word), "v"(a), "v"(b));
            if (KIND == 70) asm volatile("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(x[i]) : "v"(a), "v"(word));
            if (KIND == 71) asm volatile("v_add_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(x[i]) : "v"(a), "v"(word));
            if (KIND == 72) asm volatile("v_cvt_pk_f32_fp8 %0, %1" : "=v"(y[i & 7]) : "v"(word));
            if (KIND == 73) asm volatile("v_cvt_scalef32_pk_f32_fp8 %0, %1, %2" : "=v"(y[i & 7]) : "v"(word), "v"(a));
            if (KIND == 74) asm volatile("v_cvt_f32_fp8 %0, %1" : "=v"(x[i]) : "v"(word));
            if (KIND == 75) asm volatile("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_fma_f32 %0, %0, %1, %3\n v_fma_f32 %0, %0, %1, %3" : "=v"(x[i]) : "v"(a), "v"(word), "v"(b));
            if (KIND == 76) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(x[i]) : "v"(a));
            if (KIND == 77) asm volatile("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(x[i]) : "v"(a), "v"(word));
            if (KIND == 78) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3" : "=v"(x[i]) : "v"(word));
            if (KIND == 79) asm volatile("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(x[i]) : "v"(word), "v"(word));
        }
    }
    const unsigned long long c1 = clock64();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = (float)(mask & 1ull) + (float)y[0] + (float)y[3] + (float)y[7];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clocks[0] = c1 - c0; clocks[1] = r1 - r0; }
}

template <int KIND>
int run(const char* name, int cus)
{
    float* out; unsigned long long* clk;
    CHECK(hipMalloc((void**)&out, sizeof(float) * 256 * cus * 8));
    CHECK(hipMalloc((void**)&clk, 16));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int waves = 1; waves <= 8; waves *= 2) {          // blocks of 4 waves: one wave per SIMD per block
        const int blocks = cus * waves;
        k_rate<KIND><<<blocks, 256>>>(out, 1.0001f, 0.5f, 0x01020304u, clk);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        k_rate<KIND><<<blocks, 256>>>(out, 1.0001f, 0.5f, 0x01020304u, clk);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[2]; CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
        const double instr_per_simd = (double)kIters * 16.0 * waves;      // wave instructions issued on one SIMD
        const double mhz = (double)h[0] / ((double)h[1] / 100.0);         // shader clocks per microsecond (the realtime counter runs at 100 MHz)
        printf("%-18s waves/SIMD %d: %.3f ms, wave-instr/cycle/SIMD %.3f by clock64 (%.0f clocks, %.0f MHz), %.3f by event at 2.4 GHz\n",
               name, waves, ms, instr_per_simd / (double)h[0], (double)h[0], mhz, instr_per_simd / (ms * 1e-3 * 2.4e9));
    }
    return 0;
}

int main(int argc, char** argv)
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs\n", p.gcnArchName, cus);
    if (argc > 1 && argv[1][0] == 's') {       // sub-dword (SDWA) operand forms, fp8 conversions
        if (run<0>("v_fma_f32", cus)) return 1;
        if (run<23>("v_or_b32", cus)) return 1;
        if (run<2>("v_cvt_f32_ubyte1", cus)) return 1;
        if (run<70>("v_or_b32_sdwa byte", cus)) return 1;
        if (run<77>("v_and_b32_sdwa byte", cus)) return 1;
        if (run<78>("v_mov_b32_sdwa byte", cus)) return 1;
        if (run<79>("v_lshlrev_b32_sdwa", cus)) return 1;
        if (run<71>("v_add_f32_sdwa word", cus)) return 1;
        if (run<72>("v_cvt_pk_f32_fp8", cus)) return 1;
        if (run<73>("v_cvt_scalef32_pk_f32_fp8", cus)) return 1;
        if (run<74>("v_cvt_f32_fp8", cus)) return 1;
        if (run<76>("v_mov_b32_dpp", cus)) return 1;
        if (run<75>("or_sdwa+fma+fma (x3 instr)", cus)) return 1;
        if (run<68>("cvt+fma+fma (x3 instr)", cus)) return 1;
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'm') {       // the mixed-precision and 16-bit forms only
        if (run<0>("v_fma_f32", cus)) return 1;
        if (run<2>("v_cvt_f32_ubyte1", cus)) return 1;
        if (run<60>("v_fma_mix_f32 lo", cus)) return 1;
        if (run<61>("v_fma_mix_f32 hi", cus)) return 1;
        if (run<69>("v_fma_mix_f32 indep", cus)) return 1;
        if (run<62>("v_cvt_f32_f16", cus)) return 1;
        if (run<63>("v_pk_fma_f16", cus)) return 1;
        if (run<64>("v_pk_max_f16", cus)) return 1;
        if (run<65>("v_pk_add_f16", cus)) return 1;
        if (run<66>("v_cvt_f32_ubyte sdwa", cus)) return 1;
        if (run<67>("mix+fma (x2 instr)", cus)) return 1;
        if (run<68>("cvt+fma+fma (x3 instr)", cus)) return 1;
        return 0;
    }
    if (run<50>("e32cnd+3add (x4 instr)", cus)) return 1;
    if (run<51>("e64cnd+3add (x4 instr)", cus)) return 1;
    if (run<52>("cmp+e32cnd (x2 instr)", cus)) return 1;
    if (run<53>("cmp+e64cnd (x2 instr)", cus)) return 1;
    if (run<40>("cndmask e64 vcc", cus)) return 1;
    if (run<41>("cndmask e32 vcc swapped", cus)) return 1;
    if (run<42>("cndmask e32 vcc indep", cus)) return 1;
    if (run<43>("cndmask e64 sgpr indep", cus)) return 1;
    if (run<44>("v_addc_co_u32 vcc", cus)) return 1;
    if (run<30>("cndmask e32 vcc", cus)) return 1;
    if (run<1>("cndmask e64 sgpr", cus)) return 1;
    if (run<0>("v_fma_f32", cus)) return 1;
    if (run<2>("v_cvt_f32_ubyte1", cus)) return 1;
    if (run<3>("v_max3_f32", cus)) return 1;
    if (run<4>("v_cmp_lt_f32 vcc", cus)) return 1;
    if (run<5>("v_mul_f32", cus)) return 1;
    if (run<6>("v_add_f32", cus)) return 1;
    if (run<7>("v_max_f32", cus)) return 1;
    if (run<8>("v_and_b32", cus)) return 1;
    if (run<9>("v_add_u32", cus)) return 1;
    if (run<10>("v_mov_b32", cus)) return 1;
    if (run<11>("v_lshl_add_u32", cus)) return 1;
    if (run<12>("v_pk_fma_f32", cus)) return 1;
    if (run<13>("v_cmp_lt_f32 sgpr", cus)) return 1;
    if (run<14>("v_min_u32", cus)) return 1;
    if (run<15>("v_max_i32", cus)) return 1;
    if (run<16>("v_min3_u32", cus)) return 1;
    if (run<17>("v_med3_f32", cus)) return 1;
    if (run<18>("v_bfi_b32", cus)) return 1;
    if (run<19>("v_perm_b32", cus)) return 1;
    if (run<20>("v_ashrrev_i32", cus)) return 1;
    if (run<21>("v_lshrrev_b32", cus)) return 1;
    if (run<22>("v_xor_b32", cus)) return 1;
    if (run<23>("v_or_b32", cus)) return 1;
    if (run<24>("v_and_or_b32", cus)) return 1;
    if (run<25>("v_min_f32", cus)) return 1;
    if (run<26>("v_sub_f32", cus)) return 1;
    if (run<27>("v_cvt_f32_u32", cus)) return 1;
    if (run<28>("v_bfe_u32", cus)) return 1;
    if (run<29>("v_cmp_lt_u32", cus)) return 1;
    if (run<31>("v_fmac_f32", cus)) return 1;
    if (run<32>("v_add3_u32", cus)) return 1;
    if (run<33>("v_lshl_or_b32", cus)) return 1;
    if (run<34>("v_sub_u32", cus)) return 1;
    if (run<35>("v_mul_lo_u32", cus)) return 1;
    if (run<36>("v_rcp_f32", cus)) return 1;
    if (run<37>("v_cvt_f32_ubyte0", cus)) return 1;
    if (run<38>("v_lshlrev_b32", cus)) return 1;
    if (run<39>("v_cmp_class", cus)) return 1;
    return 0;
}
