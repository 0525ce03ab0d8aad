// Which operand forms of v_fma_f32 / v_fmac_f32 issue at the fast FP32 rate on gfx950 (8 waves/SIMD, four independent chains)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define OP4(ASM, ...) \
    asm volatile(ASM : "+v"(a0) : __VA_ARGS__); asm volatile(ASM : "+v"(a1) : __VA_ARGS__); \
    asm volatile(ASM : "+v"(a2) : __VA_ARGS__); asm volatile(ASM : "+v"(a3) : __VA_ARGS__);
template <int MODE>
__global__ void k(float* out, int iters, float s0, float s1)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    float b = 1.0f + threadIdx.x * 1e-3f, c = 0.5f - threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) { OP4("v_fma_f32 %0, %0, %1, %2", "v"(b), "v"(c)) }
            if (MODE == 1) { OP4("v_fma_f32 %0, %1, %0, %2", "s"(s0), "v"(c)) }
            if (MODE == 2) { OP4("v_fmac_f32 %0, %1, %2", "s"(s0), "v"(c)) }
            if (MODE == 3) { OP4("v_fma_f32 %0, -%0, %1, %2", "v"(b), "v"(c)) }
            if (MODE == 4) { OP4("v_fma_f32 %0, %0, %1, %2", "v"(b), "s"(s0)) }
            if (MODE == 5) { OP4("v_fma_f32 %0, %0, %1, %2 clamp", "v"(b), "v"(c)) }
            if (MODE == 6) { OP4("v_fma_f32 %0, %0, %0, %1", "v"(c)) }
            if (MODE == 7) { OP4("v_fma_f32 %0, -%0, %0, %1", "s"(s0)) }
            if (MODE == 8) { OP4("v_fmac_f32 %0, %1, %2", "v"(b), "v"(c)) }
            if (MODE == 9) { OP4("v_mul_f32 %0, %1, %0", "s"(s0)) }
            if (MODE == 10) { OP4("v_fma_f32 %0, %1, %2, %0", "s"(s0), "v"(c)) }
            if (MODE == 11) { OP4("v_sub_f32 %0, %1, %0", "s"(s0)) }
            if (MODE == 12) { OP4("v_mad_f32 %0, %0, %1, %2", "v"(b), "v"(c)) }
            if (MODE == 13) { OP4("v_fma_f32 %0, %0, %1, %2", "v"(b), "v"(b)) }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
template <int MODE> float run(float* out, int iters, int blocks, hipEvent_t e0, hipEvent_t e1)
{
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.37f, -0.21f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}
int main()
{
    float* out; CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 1024, blocks = 256 * 8;
    const char* names[14] = { "v_fma_f32 a, a, v, v", "v_fma_f32 a, s, a, v", "v_fmac_f32 a, s, v", "v_fma_f32 a, -a, v, v", "v_fma_f32 a, a, v, s", "v_fma_f32 a, a, v, v clamp",
                              "v_fma_f32 a, a, a, v", "v_fma_f32 a, -a, a, s", "v_fmac_f32 a, v, v", "v_mul_f32 a, s, a", "v_fma_f32 a, s, v, a", "v_sub_f32 a, s, a", "v_mad_f32 a, a, v, v", "v_fma_f32 a, a, v, v (same v twice)" };
    float ms[14];
    ms[0] = run<0>(out, iters, blocks, e0, e1); ms[1] = run<1>(out, iters, blocks, e0, e1); ms[2] = run<2>(out, iters, blocks, e0, e1); ms[3] = run<3>(out, iters, blocks, e0, e1);
    ms[4] = run<4>(out, iters, blocks, e0, e1); ms[5] = run<5>(out, iters, blocks, e0, e1); ms[6] = run<6>(out, iters, blocks, e0, e1); ms[7] = run<7>(out, iters, blocks, e0, e1);
    ms[8] = run<8>(out, iters, blocks, e0, e1); ms[9] = run<9>(out, iters, blocks, e0, e1); ms[10] = run<10>(out, iters, blocks, e0, e1); ms[11] = run<11>(out, iters, blocks, e0, e1);
    ms[12] = 0; ms[13] = run<13>(out, iters, blocks, e0, e1);
    for (int m = 0; m < 14; ++m) if (m != 12) printf("%-38s %8.3f ms  %6.2f SIMD cycles per instruction\n", names[m], ms[m], ms[m] * 1e-3 * 2.4e9 / (8.0 * iters * 64));
    return 0;
}
