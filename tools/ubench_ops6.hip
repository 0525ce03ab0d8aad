// Packed FP32 (v_pk_fma_f32) with a scalar register pair as an operand on gfx950 (8 waves/SIMD): cost per instruction, and the
// plane filter written with it, cycles per ray-sphere pair:
//   C  3 x v_pk_fma_f32 (two spheres at once, records as SGPR pairs) + 2 x v_alignbit_b32
//   D  3 x v_pk_fma_f32, the third with clamp on pre-scaled operands (0.0 or 1.0 per half), + v_pk_fma_f32 acc = 2 acc + s
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
#define OP4(ASM, ...) \
    asm volatile(ASM : "+v"(a0) : __VA_ARGS__); asm volatile(ASM : "+v"(a1) : __VA_ARGS__); \
    asm volatile(ASM : "+v"(a2) : __VA_ARGS__); asm volatile(ASM : "+v"(a3) : __VA_ARGS__);
template <int MODE>
__global__ void k(float* out, int iters, f2 s0, f2 s1, f2 s2)
{
    const float t = threadIdx.x;
    f2 a0 = { t, t + 1 }, a1 = { t + 2, t + 3 }, a2 = { t + 4, t + 5 }, a3 = { t + 6, t + 7 };
    f2 b = { 1.0f + t * 1e-3f, 1.0f - t * 1e-3f }, c = { 0.5f - t * 1e-3f, 0.25f };
    f2 uz = { 0.3f + t * 1e-3f, 0.3f + t * 1e-3f }, ux = { 0.7f, 0.7f }, nou = { 0.1f, 0.1f }, acc = { 0.0f, 0.0f }, two = { 2.0f, 2.0f };
    uint32_t m0 = 0, m1 = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) { OP4("v_pk_fma_f32 %0, %0, %1, %2", "v"(b), "v"(c)) }
            if (MODE == 1) { OP4("v_pk_fma_f32 %0, %1, %0, %2", "s"(s0), "v"(c)) }
            if (MODE == 2) { OP4("v_pk_fma_f32 %0, %0, %0, %1 neg_lo:[1,0,0] neg_hi:[1,0,0]", "s"(s0)) }
            if (MODE == 3) { OP4("v_pk_fma_f32 %0, %0, %1, %2 clamp", "v"(b), "v"(c)) }
            if (MODE == 4) {
                f2 p, q;
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p) : "s"(s0), "v"(uz), "v"(nou));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p) : "s"(s1), "v"(ux));
                asm volatile("v_pk_fma_f32 %0, %1, %1, %2 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(q) : "v"(p), "s"(s2));
                asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(m0) : "v"(q.x));
                asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(m1) : "v"(q.y));
            }
            if (MODE == 5) {
                f2 p, q;
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p) : "s"(s0), "v"(uz), "v"(nou));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p) : "s"(s1), "v"(ux));
                asm volatile("v_pk_fma_f32 %0, %1, %1, %2 neg_lo:[1,0,0] neg_hi:[1,0,0] clamp" : "=v"(q) : "v"(p), "s"(s2));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(two), "v"(q));
            }
        }
        if (MODE == 5) { acc.x *= 1.52587890625e-05f; acc.y *= 1.52587890625e-05f; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a1.y + a2.x + a3.y + acc.x + acc.y + (float)(m0 ^ m1);
}
template <int MODE> float run(float* out, int iters, int blocks, hipEvent_t e0, hipEvent_t e1)
{
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, f2{ 0.37f, 0.35f }, f2{ -0.21f, -0.2f }, f2{ 0.04f, 0.05f });
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
    const char* names[6] = { "v_pk_fma_f32 a, a, v, v", "v_pk_fma_f32 a, s, a, v", "v_pk_fma_f32 a, -a, a, s", "v_pk_fma_f32 a, a, v, v clamp",
                             "two pairs: 3 pk_fma (s) + 2 alignbit", "two pairs: 3 pk_fma (s, clamp) + pk_fma acc" };
    const double per_iter[6] = { 64, 64, 64, 64, 32, 32 };     // instructions, or ray-sphere pairs, per loop iteration
    float ms[6];
    ms[0] = run<0>(out, iters, blocks, e0, e1); ms[1] = run<1>(out, iters, blocks, e0, e1); ms[2] = run<2>(out, iters, blocks, e0, e1);
    ms[3] = run<3>(out, iters, blocks, e0, e1); ms[4] = run<4>(out, iters, blocks, e0, e1); ms[5] = run<5>(out, iters, blocks, e0, e1);
    for (int m = 0; m < 6; ++m) printf("%-44s %8.3f ms  %6.2f SIMD cycles per %s\n", names[m], ms[m], ms[m] * 1e-3 * 2.4e9 / (8.0 * iters * per_iter[m]), m < 4 ? "instruction" : "ray-sphere pair");
    return 0;
}
