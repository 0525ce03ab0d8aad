// Per-instruction issue cost on gfx950 at 8 waves/SIMD, second set: packed half, dot, bit-field and 3-operand integer ops.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define OP8(ASM) \
    asm volatile(ASM : "+v"(a0) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a1) : "v"(b), "v"(c)); \
    asm volatile(ASM : "+v"(a2) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a3) : "v"(b), "v"(c)); \
    asm volatile(ASM : "+v"(a4) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a5) : "v"(b), "v"(c)); \
    asm volatile(ASM : "+v"(a6) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a7) : "v"(b), "v"(c));
template <int MODE>
__global__ void k(uint32_t* out, int iters, uint32_t bb, uint32_t cc)
{
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = bb + threadIdx.x, c = cc;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) { OP8("v_pk_fma_f16 %0, %0, %1, %2") }
            if (MODE == 1) { OP8("v_pk_mul_f16 %0, %0, %1") }
            if (MODE == 2) { OP8("v_pk_add_f16 %0, %0, %1") }
            if (MODE == 3) { OP8("v_dot2_f32_f16 %0, %1, %2, %0") }
            if (MODE == 4) { OP8("v_dot2c_f32_f16 %0, %1, %2") }
            if (MODE == 5) { OP8("v_bfi_b32 %0, %1, %2, %0") }
            if (MODE == 6) { OP8("v_perm_b32 %0, %0, %1, %2") }
            if (MODE == 7) { OP8("v_lshl_add_u32 %0, %0, 1, %1") }
            if (MODE == 8) { OP8("v_and_b32 %0, %0, %1") }
            if (MODE == 9) { OP8("v_or_b32 %0, %0, %1") }
            if (MODE == 10) { OP8("v_lshlrev_b32 %0, 1, %0") }
            if (MODE == 11) { OP8("v_fma_mix_f32 %0, %0, %1, %2") }
            if (MODE == 12) { OP8("v_fma_f32 %0, %0, %1, %2") }
            if (MODE == 13) { OP8("v_min3_f32 %0, %0, %1, %2") }
            if (MODE == 14) { OP8("v_med3_f32 %0, %0, %1, %2") }
            if (MODE == 15) { OP8("v_add3_u32 %0, %0, %1, %2") }
            if (MODE == 16) { OP8("v_xad_u32 %0, %0, %1, %2") }
            if (MODE == 17) { OP8("v_or3_b32 %0, %0, %1, %2") }
            if (MODE == 18) { OP8("v_fma_f16 %0, %0, %1, %2") }
            if (MODE == 19) { OP8("v_cvt_f16_f32 %0, %0") }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
template <int MODE> float run(uint32_t* out, int iters, int blocks, hipEvent_t e0, hipEvent_t e1)
{
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 0x3f7ff000u, 0x3a800000u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}
int main()
{
    uint32_t* out; CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 1024, blocks = 256 * 8;
    const char* names[20] = {"v_pk_fma_f16", "v_pk_mul_f16", "v_pk_add_f16", "v_dot2_f32_f16", "v_dot2c_f32_f16", "v_bfi_b32", "v_perm_b32", "v_lshl_add_u32", "v_and_b32", "v_or_b32", "v_lshlrev_b32", "v_fma_mix_f32", "v_fma_f32 (ref)", "v_min3_f32", "v_med3_f32", "v_add3_u32", "v_xad_u32", "v_or3_b32", "v_fma_f16", "v_cvt_f16_f32"};
    float ms[20];
    ms[0] = run<0>(out, iters, blocks, e0, e1);
    ms[1] = run<1>(out, iters, blocks, e0, e1);
    ms[2] = run<2>(out, iters, blocks, e0, e1);
    ms[3] = run<3>(out, iters, blocks, e0, e1);
    ms[4] = run<4>(out, iters, blocks, e0, e1);
    ms[5] = run<5>(out, iters, blocks, e0, e1);
    ms[6] = run<6>(out, iters, blocks, e0, e1);
    ms[7] = run<7>(out, iters, blocks, e0, e1);
    ms[8] = run<8>(out, iters, blocks, e0, e1);
    ms[9] = run<9>(out, iters, blocks, e0, e1);
    ms[10] = run<10>(out, iters, blocks, e0, e1);
    ms[11] = run<11>(out, iters, blocks, e0, e1);
    ms[12] = run<12>(out, iters, blocks, e0, e1);
    ms[13] = run<13>(out, iters, blocks, e0, e1);
    ms[14] = run<14>(out, iters, blocks, e0, e1);
    ms[15] = run<15>(out, iters, blocks, e0, e1);
    ms[16] = run<16>(out, iters, blocks, e0, e1);
    ms[17] = run<17>(out, iters, blocks, e0, e1);
    ms[18] = run<18>(out, iters, blocks, e0, e1);
    ms[19] = run<19>(out, iters, blocks, e0, e1);
    for (int m = 0; m < 20; ++m) {
        double ops = (double)blocks * 4 * iters * 64;
        printf("%-16s %.3f ms  %.2f cycles/instr\n", names[m], ms[m], ms[m] * 1e-3 * 2.4e9 * 1024 / ops);
    }
    return 0;
}
