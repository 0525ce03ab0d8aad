// Two ways of collecting the plane filter's verdicts on gfx950 at 8 waves/SIMD, cycles per ray-sphere pair:
//   A  3 x v_fma_f32 + v_alignbit_b32 (sign bit appended to an integer mask)
//   B  3 x v_fma_f32, the third with the clamp modifier on pre-scaled operands (result exactly 0.0 or 1.0), + v_fma_f32 acc = 2 acc + s
// and the two new instruction forms on their own.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void k(float* out, int iters, float sx, float sz, float sr)
{
    float ux = 1.0f + threadIdx.x * 1e-3f, uz = 0.5f - threadIdx.x * 1e-3f, nou = 0.25f;
    float accA = 0.0f, accB = 0.0f; uint32_t mask = 0;
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) {
                float pu, t;
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(pu) : "s"(sz), "v"(uz), "v"(nou));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(pu) : "s"(sx), "v"(ux));
                asm volatile("v_fma_f32 %0, -%1, %1, %2" : "=v"(t) : "v"(pu), "s"(sr));
                asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(mask) : "v"(t));
            }
            if (MODE == 1) {
                float pu, t;
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(pu) : "s"(sz), "v"(uz), "v"(nou));
                asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(pu) : "s"(sx), "v"(ux));
                asm volatile("v_fma_f32 %0, -%1, %1, %2 clamp" : "=v"(t) : "v"(pu), "s"(sr));
                if (u & 1) asm volatile("v_fma_f32 %0, %0, 2.0, %1" : "+v"(accA) : "v"(t));
                else       asm volatile("v_fma_f32 %0, %0, 2.0, %1" : "+v"(accB) : "v"(t));
            }
            if (MODE == 2) {
                asm volatile("v_fma_f32 %0, -%0, %0, %1 clamp" : "+v"(a0) : "s"(sr)); asm volatile("v_fma_f32 %0, -%0, %0, %1 clamp" : "+v"(a1) : "s"(sr));
                asm volatile("v_fma_f32 %0, -%0, %0, %1 clamp" : "+v"(a2) : "s"(sr)); asm volatile("v_fma_f32 %0, -%0, %0, %1 clamp" : "+v"(a3) : "s"(sr));
            }
            if (MODE == 3) {
                asm volatile("v_fma_f32 %0, %0, 2.0, %1" : "+v"(a0) : "v"(ux)); asm volatile("v_fma_f32 %0, %0, 2.0, %1" : "+v"(a1) : "v"(ux));
                asm volatile("v_fma_f32 %0, %0, 2.0, %1" : "+v"(a2) : "v"(ux)); asm volatile("v_fma_f32 %0, %0, 2.0, %1" : "+v"(a3) : "v"(ux));
            }
        }
        if (MODE == 1) { accA = accA * 1.52587890625e-05f; accB = accB * 1.52587890625e-05f; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = accA + accB + (float)mask + a0 + a1 + a2 + a3;
}
template <int MODE> float run(float* out, int iters, int blocks, hipEvent_t e0, hipEvent_t e1)
{
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.37f, -0.21f, 0.04f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}
int main()
{
    float* out; CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 2048, blocks = 256 * 8;
    const char* names[4] = { "pair: 3 fma + alignbit", "pair: 2 fma + fma clamp + fma acc", "v_fma_f32 -a, a, s clamp", "v_fma_f32 a, 2.0, v" };
    const double per_iter[4] = { 16, 16, 64, 64 };
    float ms[4];
    ms[0] = run<0>(out, iters, blocks, e0, e1); ms[1] = run<1>(out, iters, blocks, e0, e1); ms[2] = run<2>(out, iters, blocks, e0, e1); ms[3] = run<3>(out, iters, blocks, e0, e1);
    for (int m = 0; m < 4; ++m) printf("%-36s %8.3f ms  %6.2f SIMD cycles per unit\n", names[m], ms[m], ms[m] * 1e-3 * 2.4e9 / (8.0 * iters * per_iter[m]));
    return 0;
}
