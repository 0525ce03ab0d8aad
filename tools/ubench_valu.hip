// VALU issue-rate microbenchmark for gfx950: plain v_fma_f32 vs packed v_pk_fma_f32,
// at 1/2/4/8 waves per SIMD.  Prints achieved TFLOP/s per variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_fma(float* out, int iters, float m, float c) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
            a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

__global__ void k_pkfma(float* out, int iters, float m, float c) {
    v2f mm = {m, m}, cc = {c, c};
    v2f a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a0 = __builtin_elementwise_fma(a0, mm, cc); a1 = __builtin_elementwise_fma(a1, mm, cc);
            a2 = __builtin_elementwise_fma(a2, mm, cc); a3 = __builtin_elementwise_fma(a3, mm, cc);
            a4 = __builtin_elementwise_fma(a4, mm, cc); a5 = __builtin_elementwise_fma(a5, mm, cc);
            a6 = __builtin_elementwise_fma(a6, mm, cc); a7 = __builtin_elementwise_fma(a7, mm, cc);
        }
    }
    v2f s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

// mixed: sub + fma with an SGPR operand per op, like the sphere sweep
__global__ void k_mix(float* out, int iters, const float* __restrict__ sp) {
    float ox = threadIdx.x * 0.01f, oy = ox + 1, oz = ox + 2, dx = 0.3f + ox, dy = 0.5f, dz = 0.7f;
    float a = dx * dx + dy * dy + dz * dz;
    float best = 1e30f;
    for (int i = 0; i < iters; ++i) {
        const float4 s = reinterpret_cast<const float4*>(sp)[i & 1023];
        float x = ox - s.x, y = oy - s.y, z = oz - s.z;
        float b = x * dx; b = __builtin_fmaf(y, dy, b); b = __builtin_fmaf(z, dz, b);
        float c = __builtin_fmaf(x, x, s.w); c = __builtin_fmaf(y, y, c); c = __builtin_fmaf(z, z, c);
        float d = __builtin_fmaf(b, b, -(a * c));
        best = fminf(best, d);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = best;
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    float* out; CHECK(hipMalloc(&out, 256 * 8 * 256 * 4 * sizeof(float)));
    float* sp; CHECK(hipMalloc(&sp, 1024 * 16));
    std::vector<float> h(4096); for (int i = 0; i < 4096; ++i) h[i] = (i % 4 == 3) ? -0.04f : (float)(i % 37) * 0.1f;
    CHECK(hipMemcpy(sp, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 8192;
    for (int wps = 1; wps <= 8; wps *= 2) {
        int blocks = p.multiProcessorCount * wps;  // 256 threads = 4 waves = 1 wave/SIMD per block
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                CHECK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, iters, 0.999f, 0.001f);
                else if (mode == 1) hipLaunchKernelGGL(k_pkfma, dim3(blocks), dim3(256), 0, 0, out, iters, 0.999f, 0.001f);
                else hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(256), 0, 0, out, iters * 4, sp);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            double flops;
            if (mode == 0) flops = (double)blocks * 256 * iters * 32 * 2;
            else if (mode == 1) flops = (double)blocks * 256 * iters * 32 * 4;
            else flops = (double)blocks * 256 * iters * 4 * 18;
            printf("waves/SIMD %d  %-8s  %.3f ms  %.1f TFLOP/s\n", wps, mode == 0 ? "fma" : mode == 1 ? "pk_fma" : "sweepmix", best, flops / best / 1e9);
        }
    }
    return 0;
}
