// The plane filter of the render kernel in isolation at the render kernel's occupancy (512-thread blocks, 3 per CU = 6 waves
// per SIMD): SIMD cycles per ray-sphere pair for the scalar-operand form and the packed form, with the records fetched by
// scalar loads as in the kernel.  488 spheres, every wave sweeps them `trips` times.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(4))) f4v* Rec4;
typedef const __attribute__((address_space(4))) f16v* Rec16;

template <int MODE>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(6, 6))) k(const float* recs, uint32_t* out, int n_pad, int trips)
{
    extern __shared__ uint32_t lds[];                      // 51 KB per block, as in the render kernel: 3 blocks per CU
    uint32_t* words = lds + (threadIdx.x >> 6) * 16 * 64 + (threadIdx.x & 63);
    float ux = 0.6f + threadIdx.x * 1e-4f, uz = 0.8f - threadIdx.x * 1e-4f, nou = -0.3f;
    uint32_t acc = 0;
    for (int t = 0; t < trips; ++t) {
        nou += 1e-3f;
        int w = 0;
        for (int base = 0; base + 32 <= n_pad; base += 32, ++w) {
            uint32_t mask = 0;
            if (MODE == 0) {
                const Rec4 r = (Rec4)recs;
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    const f4v f = r[base + j];
                    const float pu = fmaf(f.x, ux, fmaf(f.y, uz, nou));
                    mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(fmaf(-pu, pu, f.z)), 31);
                }
            } else if (MODE == 1) {
                const Rec16 c = (Rec16)recs + (base >> 5) * 6;
                const f2v zx = { uz, ux }, nn = { nou, nou };
                uint32_t m_lo = 0, m_hi = 0;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const f16v cx16 = c[g], cz16 = c[2 + g], r16 = c[4 + g];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const f2v cx2 = { cx16[2 * i], cx16[2 * i + 1] }, cz2 = { cz16[2 * i], cz16[2 * i + 1] }, r2 = { r16[2 * i], r16[2 * i + 1] };
                        f2v pu, sg;
                        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(pu) : "s"(cz2), "v"(zx), "v"(nn));
                        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(pu) : "s"(cx2), "v"(zx), "v"(pu));
                        asm("v_pk_fma_f32 %0, %1, %1, %2 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(sg) : "v"(pu), "s"(r2));
                        m_lo = __builtin_amdgcn_alignbit(m_lo, __float_as_uint(sg.x), 31);
                        m_hi = __builtin_amdgcn_alignbit(m_hi, __float_as_uint(sg.y), 31);
                    }
                }
                mask = (m_lo << 16) | m_hi;
            }
            if (MODE == 2) {
                // pairs (i, i + 16) of the chunk; per pair 8 floats (cx_i, cx_j, cz_i, cz_j, R_i, R_j, -, -), two pairs per 16-float load
                const Rec16 c = (Rec16)recs + (base >> 5) * 8;
                const f2v zx = { uz, ux }, nn = { nou, nou };
                uint32_t m_lo = 0, m_hi = 0;
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const f16v q = c[g];
                    asm volatile("" :: "s"(q));          // the whole 16-register tuple is wanted: one s_load_dwordx16
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const f2v cx2 = { q[8 * i], q[8 * i + 1] }, cz2 = { q[8 * i + 2], q[8 * i + 3] }, r2 = { q[8 * i + 4], q[8 * i + 5] };
                        f2v pu, sg;
                        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(pu) : "s"(cz2), "v"(zx), "v"(nn));
                        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(pu) : "s"(cx2), "v"(zx), "v"(pu));
                        asm("v_pk_fma_f32 %0, %1, %1, %2 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(sg) : "v"(pu), "s"(r2));
                        m_lo = __builtin_amdgcn_alignbit(m_lo, __float_as_uint(sg.x), 31);
                        m_hi = __builtin_amdgcn_alignbit(m_hi, __float_as_uint(sg.y), 31);
                    }
                }
                mask = (m_lo << 16) | m_hi;
            }
            words[w * 64] = mask;
            acc ^= mask;
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}
template <int MODE> float run(const float* recs, uint32_t* out, int n_pad, int trips, hipEvent_t e0, hipEvent_t e1)
{
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256 * 3), dim3(512), 52248, 0, recs, out, n_pad, trips);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}
int main()
{
    const int n_pad = 480;          // 15 chunks of 32
    std::vector<float> h(6 * 16 * 16 + 4 * 512, 0.0f);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.01f * (float)((i * 2654435761u) % 1000) - 5.0f;
    float* recs; CHECK(hipMalloc(&recs, h.size() * 4)); CHECK(hipMemcpy(recs, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    uint32_t* out; CHECK(hipMalloc(&out, 256 * 3 * 512 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int trips = 2000;
    const float a = run<0>(recs, out, n_pad, trips, e0, e1), b = run<1>(recs, out, n_pad, trips, e0, e1), c = run<2>(recs, out, n_pad, trips, e0, e1);
    // 6 waves per SIMD share it: SIMD cycles per pair = time x clock / (waves per SIMD x trips x spheres)
    printf("scalar-operand form   %8.3f ms  %6.2f SIMD cycles per ray-sphere pair\n", a, a * 1e-3 * 2.4e9 / (6.0 * trips * n_pad));
    printf("packed form           %8.3f ms  %6.2f SIMD cycles per ray-sphere pair\n", b, b * 1e-3 * 2.4e9 / (6.0 * trips * n_pad));
    printf("packed, 8 floats/pair %8.3f ms  %6.2f SIMD cycles per ray-sphere pair\n", c, c * 1e-3 * 2.4e9 / (6.0 * trips * n_pad));
    return 0;
}
