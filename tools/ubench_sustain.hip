// Sustained FP32 FMA rate against run length on gfx950: the same dependent-chain FMA kernel (8 chains per lane, 8 waves
// per SIMD) run for about 1, 10, 100 and 1000 ms.  Rate / (256 CU x 4 SIMD x 32 lanes x 2 flop) = effective shader clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void __launch_bounds__(256) k_fma(float* out, long iters, float m, float c)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (long i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
            a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main()
{
    const int blocks = 256 * 8;          // 8 blocks of 4 waves per CU = 8 waves per SIMD
    float* out; CHECK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(float)));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, 1000L, 0.999f, 0.001f);
    CHECK(hipDeviceSynchronize());
    for (long iters : { 2000L, 20000L, 200000L, 2000000L, 2000L }) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, iters, 0.999f, 0.001f);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double flop = (double)blocks * 256 * iters * 64 * 2;
        const double tf = flop / (ms * 1e-3) / 1e12;
        printf("%9ld iterations  %9.2f ms  %6.1f TFLOP/s  -> effective clock %.2f GHz\n", iters, ms, tf, tf * 1e12 / (256.0 * 4 * 32 * 2) / 1e9);
    }
    // the same work as ten 10 ms launches enqueued back to back (no host wait in between), and with a wait after each
    for (int wait_each = 0; wait_each < 2; ++wait_each) {
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int k = 0; k < 10; ++k) {
            hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, 20000L, 0.999f, 0.001f);
            if (wait_each) CHECK(hipDeviceSynchronize());
        }
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double tf = (double)blocks * 256 * 20000L * 64 * 2 * 10 / (ms * 1e-3) / 1e12;
        printf("10 x 20000 iterations, %s: %9.2f ms  %6.1f TFLOP/s\n", wait_each ? "host waits after each launch" : "enqueued back to back", ms, tf);
    }
    return 0;
}
