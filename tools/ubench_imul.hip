// Integer multiply issue rates on gfx950: v_mul_lo_u32 / v_mul_hi_u32 / v_mad_u64_u32, relative to v_fma_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int MODE>
__global__ void k(uint32_t* out, int iters, uint32_t m)
{
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) a[i] = a[i] * m;
            else if (MODE == 1) a[i] = __umulhi(a[i], m) ^ (uint32_t)it;
            else if (MODE == 2) { uint64_t p; uint64_t c; asm volatile("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p), "=s"(c) : "v"(a[i]), "v"(m)); a[i] = (uint32_t)p ^ (uint32_t)(p >> 32); }
            else if (MODE == 3) { a[i] = (a[i] ^ m) + (uint32_t)it; }
        }
    }
    uint32_t s = 0; for (int i = 0; i < 8; ++i) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    uint32_t* out; CHECK(hipMalloc(&out, 256 * 8 * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 4096, blocks = 256 * 8;
    const char* names[4] = {"v_mul_lo_u32", "v_mul_hi_u32(+xor)", "v_mad_u64_u32(+xor)", "xor+add"};
    for (int mode = 0; mode < 4; ++mode) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 0xD2511F53u);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 0xD2511F53u);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 0xD2511F53u);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, iters, 0xD2511F53u);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        double ops = (double)blocks * 256 / 64 * iters * 8;   // wave-instructions of the op under test
        double cyc = best * 1e-3 * 2.4e9 * 1024 / ops;       // SIMD-cycles per wave-instruction (incl. the xor where present)
        printf("%-22s %.3f ms  %.2f SIMD-cycles per loop op\n", names[mode], best, cyc);
    }
    return 0;
}
