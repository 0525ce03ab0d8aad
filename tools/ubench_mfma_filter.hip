// The plane filter of the render kernel in isolation, three ways, at the render kernel's occupancy (512-thread blocks, 3 per CU
// = 6 waves per SIMD):
//   mode 0  normalised records as scalar operands: q = cx' ux + cz' uz + nc (two FMAs with an SGPR operand), verdict = clamp(T - q q),
//           acc = 2 acc + verdict -- the shipped form;
//   mode 1  v_mfma_f32_16x16x4_f32: q[sphere i][ray j] = (cx', cz', 1/rho, 0)_i . (ux, uz, -o.u, 0)_j for 16 spheres x 16 rays per
//           instruction, records arriving one float per lane by an ordinary vector load; the two all-VGPR operations per pair stay on
//           the vector ALU; every lane ends up with 8 verdict bits per 32 spheres for each of four rays and stores them as bytes into
//           the mask words of those rays in LDS;
//   mode 2  v_mfma_f32_32x32x2_f32 with the per-ray constant in the C operand (32 registers held for the two half-wave C tiles), at
//           4 waves per SIMD because of those registers.
// Reported: SIMD cycles per (ray, sphere) pair = time x clock x SIMDs / pairs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(4))) f4v* Rec4;

__device__ __forceinline__ float verdict_of(float q, float T)
{
    // the compiler folds the [0, 1] clamp into the FMA's clamp modifier and, unlike with inline assembly, sees the read of an
    // MFMA result and keeps the wait states that read needs
    return __builtin_amdgcn_fmed3f(fmaf(-q, q, T), 0.0f, 1.0f);
}

template <int MODE, int WAVES>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) k(const float* recs, uint32_t* out, int chunks, int trips)
{
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* words = lds + wave * 16 * 64 + lane;
    uint8_t* wbytes = reinterpret_cast<uint8_t*>(lds + wave * 16 * 64);
    float ux = (0.6f + threadIdx.x * 1e-4f) * 0x1p44f, uz = (0.8f - threadIdx.x * 1e-4f) * 0x1p44f, nou = -0.3f * 0x1p44f;
    float T = 0x1p88f;
    asm volatile("" : "+v"(T));          // held in a vector register: an FMA with a scalar operand issues at half the rate
    uint32_t sum = 0;
    if (MODE == 6) { for (int i = threadIdx.x; i < chunks * 96; i += 512) lds[8 * 16 * 64 + i] = __float_as_uint(recs[i]); __syncthreads(); }
    for (int t = 0; t < trips; ++t) {
        nou += 0x1p30f;
        if (MODE == 0) {
            const Rec4 r = (Rec4)recs;
            for (int c = 0; c < chunks; ++c) {
                const float nc = nou * r[c * 32].z;
                float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    const f4v f = r[c * 32 + j];
                    const float q = fmaf(f.x, ux, fmaf(f.y, uz, nc));
                    const float v = verdict_of(q, T);
                    if (j < 16) acc0 = fmaf(acc0, 2.0f, v); else acc1 = fmaf(acc1, 2.0f, v);
                }
                const uint32_t m = ((uint32_t)acc0 << 16) | (uint32_t)acc1;
                words[c * 64] = m; sum ^= m;
            }
        } else if (MODE == 1) {
            // B operand of ray group g: lanes 0-15 ux, 16-31 uz, 32-47 -o.u, 48-63 zero, each of ray 16 g + (lane & 15); built through LDS
            words[0] = __float_as_uint(ux); words[64] = __float_as_uint(uz); words[128] = __float_as_uint(nou); words[192] = 0u;
            float B[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) B[g] = __uint_as_float(words[(lane >> 4) * 64 + 16 * g + (lane & 15) - lane]);
            const int m16 = lane & 15, q4 = lane >> 4;
            const f4v zero = { 0.0f, 0.0f, 0.0f, 0.0f };
            float an[4];
#pragma unroll
            for (int hh = 0; hh < 4; ++hh) an[hh] = recs[hh * 64 + lane];
            for (int c = 0; c < chunks; c += 2) {
                float acc[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
                float a[4];
#pragma unroll
                for (int hh = 0; hh < 4; ++hh) a[hh] = an[hh];
                // the records of the next two chunks are requested before this pair is worked on
#pragma unroll
                for (int hh = 0; hh < 4; ++hh) an[hh] = recs[((c + 2) * 2 + hh) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int hh = 0; hh < 4; ++hh) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f4v d = __builtin_amdgcn_mfma_f32_16x16x4f32(a[hh], B[g], zero, 0, 0, 0);
#pragma unroll
                        for (int v = 0; v < 4; ++v) acc[g] = fmaf(acc[g], 2.0f, verdict_of(d[v], T));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint32_t bits = (uint32_t)acc[g];
                    uint8_t* p = wbytes + ((size_t)(c * 64 + 16 * g + m16) * 4 + q4);
                    p[0] = (uint8_t)(bits >> 8); p[256] = (uint8_t)bits;
                    sum ^= bits;
                }
            }
            for (int c = 0; c < chunks; ++c) sum ^= words[c * 64];
        } else if (MODE == 3 || MODE == 4) {
            // diagnostics: the MFMAs alone, accumulating in place (3), and with the same number of vector operations on registers
            // the MFMAs do not touch (4)
            float B[4] = { ux, uz, nou, ux + uz };
            f4v d[4] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
            float acc[4] = { ux, uz, nou, 1.0f };
            for (int c = 0; c < chunks; c += 2) {
#pragma unroll
                for (int hh = 0; hh < 4; ++hh) {
                    const float a = recs[(c * 2 + hh) * 64 + lane];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        d[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, B[g], d[g], 0, 0, 0);
                        if (MODE == 4) {
#pragma unroll
                            for (int v = 0; v < 4; ++v) acc[g] = fmaf(acc[g], 2.0f, verdict_of(acc[(g + v) & 3], T));
                        }
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) sum ^= __float_as_uint(d[g][0] + d[g][1] + d[g][2] + d[g][3] + acc[g]);
        } else if (MODE == 5) {
            // v_mfma_f32_32x32x2_f32 with C = 0 and the per-ray constant added on the vector ALU: three vector operations per pair,
            // no registers held for C
            const float uxA = __shfl(ux, lane & 31), uzA = __shfl(uz, lane & 31), uxB = __shfl(ux, 32 + (lane & 31)), uzB = __shfl(uz, 32 + (lane & 31));
            const float BA = lane < 32 ? uxA : uzA, BB = lane < 32 ? uxB : uzB;
            const float ncA = __shfl(nou, lane & 31), ncB = __shfl(nou, 32 + (lane & 31));
            f16v Z;
#pragma unroll
            for (int v = 0; v < 16; ++v) Z[v] = 0.0f;
            float a = recs[lane];
            for (int c = 0; c < chunks; ++c) {
                const float a_now = a;
                a = recs[(c + 1) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);
                const f16v dA = __builtin_amdgcn_mfma_f32_32x32x2f32(a_now, BA, Z, 0, 0, 0);
                float accA = 0.0f, accA2 = 0.0f, accB = 0.0f, accB2 = 0.0f;
#pragma unroll
                for (int v = 0; v < 8; ++v) { accA = fmaf(accA, 2.0f, verdict_of(dA[v] + ncA, T)); accA2 = fmaf(accA2, 2.0f, verdict_of(dA[v + 8] + ncA, T)); }
                const f16v dB = __builtin_amdgcn_mfma_f32_32x32x2f32(a_now, BB, Z, 0, 0, 0);
#pragma unroll
                for (int v = 0; v < 8; ++v) { accB = fmaf(accB, 2.0f, verdict_of(dB[v] + ncB, T)); accB2 = fmaf(accB2, 2.0f, verdict_of(dB[v + 8] + ncB, T)); }
                __builtin_amdgcn_sched_barrier(0);
                const uint32_t a16 = ((uint32_t)accA << 8) | (uint32_t)accA2, b16 = ((uint32_t)accB << 8) | (uint32_t)accB2;
                const uint32_t other = (uint32_t)__shfl_xor((int)(lane < 32 ? b16 : a16), 32);
                const uint32_t m = lane < 32 ? (a16 << 16) | other : (other << 16) | b16;
                words[c * 64] = m; sum ^= m;
            }
        } else if (MODE == 6) {
            // records in LDS, four spheres per quad of lanes, handed to the FMAs by the quad_perm broadcast of DPP: all operands are
            // vector registers.  q = cx' ux + cz' uz + (1/rho)(-o.u): a multiply and two multiply-adds whose first operand is the
            // quad broadcast, then the same two operations as the other forms: five per pair
            const float* lrec = reinterpret_cast<const float*>(lds + 8 * 16 * 64);      // [chunk][8 groups][3 values][4 spheres]
            for (int c = 0; c < chunks; ++c) {
                float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll
                for (int g4 = 0; g4 < 8; ++g4) {
                    const float* p = lrec + ((c * 8 + g4) * 3) * 4 + (lane & 3);
                    const float cx4 = p[0], cz4 = p[4], ir4 = p[8];
#define RT_PAIR(K) { float q; \
                    asm("v_mul_f32_dpp %0, %1, %2 quad_perm:[" #K "," #K "," #K "," #K "] row_mask:0xf bank_mask:0xf" : "=v"(q) : "v"(cx4), "v"(ux)); \
                    asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[" #K "," #K "," #K "," #K "] row_mask:0xf bank_mask:0xf" : "+v"(q) : "v"(cz4), "v"(uz)); \
                    asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[" #K "," #K "," #K "," #K "] row_mask:0xf bank_mask:0xf" : "+v"(q) : "v"(ir4), "v"(nou)); \
                    const float vd = verdict_of(q, T); if (g4 < 4) acc0 = fmaf(acc0, 2.0f, vd); else acc1 = fmaf(acc1, 2.0f, vd); }
                    RT_PAIR(0) RT_PAIR(1) RT_PAIR(2) RT_PAIR(3)
                }
                const uint32_t m = ((uint32_t)acc0 << 16) | (uint32_t)acc1;
                words[c * 64] = m; sum ^= m;
            }
        } else {
            const float uxA = __shfl(ux, lane & 31), uzA = __shfl(uz, lane & 31), uxB = __shfl(ux, 32 + (lane & 31)), uzB = __shfl(uz, 32 + (lane & 31));
            const float BA = lane < 32 ? uxA : uzA, BB = lane < 32 ? uxB : uzB;
            const float ncA = __shfl(nou, lane & 31), ncB = __shfl(nou, 32 + (lane & 31));
            f16v CA, CB;
#pragma unroll
            for (int v = 0; v < 16; ++v) { CA[v] = ncA; CB[v] = ncB; }
            for (int c = 0; c < chunks; ++c) {
                const float a = recs[c * 64 + lane];
                const f16v dA = __builtin_amdgcn_mfma_f32_32x32x2f32(a, BA, CA, 0, 0, 0);
                const f16v dB = __builtin_amdgcn_mfma_f32_32x32x2f32(a, BB, CB, 0, 0, 0);
                float accA = 0.0f, accB = 0.0f;
#pragma unroll
                for (int v = 0; v < 16; ++v) accA = fmaf(accA, 2.0f, verdict_of(dA[v], T));
#pragma unroll
                for (int v = 0; v < 16; ++v) accB = fmaf(accB, 2.0f, verdict_of(dB[v], T));
                const uint32_t a16 = (uint32_t)accA, b16 = (uint32_t)accB;
                const uint32_t other = (uint32_t)__shfl_xor((int)(lane < 32 ? b16 : a16), 32);
                const uint32_t m = lane < 32 ? (a16 << 16) | other : (other << 16) | b16;
                words[c * 64] = m; sum ^= m;
            }
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = sum;
}

template <int MODE, int WAVES> float run(const float* recs, uint32_t* out, int chunks, int trips, hipEvent_t e0, hipEvent_t e1)
{
    float best = 1e30f;
    const int blocks_per_cu = WAVES / 2;
    const size_t lds = WAVES == 6 ? 52248 : 65536;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, WAVES>), dim3(256 * blocks_per_cu), dim3(512), lds, 0, recs, out, chunks, trips);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    const int chunks = 14;
    std::vector<float> h(4 * 32 * 16 + 1024, 0.0f);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.01f * (float)((i * 2654435761u) % 1000) - 5.0f;
    float* recs; CHECK(hipMalloc(&recs, h.size() * 4)); CHECK(hipMemcpy(recs, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    uint32_t* out; CHECK(hipMalloc(&out, 256 * 3 * 512 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int trips = 2000;
    const float a = run<0, 6>(recs, out, chunks, trips, e0, e1);
    const float b = run<1, 6>(recs, out, chunks, trips, e0, e1);
    const float c = run<2, 4>(recs, out, chunks, trips, e0, e1);
    CHECK(hipGetLastError());
    const double pairs_per_wave = (double)trips * chunks * 32;       // per ray; a wave does 64 of these per "wave-pair"
    printf("scalar-operand form, 6 waves/SIMD   %8.3f ms  %6.2f SIMD cycles per wave-pair (64 ray-sphere pairs)\n", a, a * 1e-3 * 2.4e9 / (6.0 * pairs_per_wave));
    printf("mfma 16x16x4 + 2 VALU, 6 waves/SIMD %8.3f ms  %6.2f\n", b, b * 1e-3 * 2.4e9 / (6.0 * pairs_per_wave));
    const float d3 = run<3, 6>(recs, out, chunks, trips, e0, e1), d4 = run<4, 6>(recs, out, chunks, trips, e0, e1);
    printf("mfma 16x16x4 alone, 6 waves/SIMD    %8.3f ms  %6.2f\n", d3, d3 * 1e-3 * 2.4e9 / (6.0 * pairs_per_wave));
    printf("mfma 16x16x4 + independent VALU     %8.3f ms  %6.2f\n", d4, d4 * 1e-3 * 2.4e9 / (6.0 * pairs_per_wave));
    const float d5 = run<5, 6>(recs, out, chunks, trips, e0, e1), d6 = run<6, 6>(recs, out, chunks, trips, e0, e1);
    printf("mfma 32x32x2, C = 0, 3 VALU, 6 w/S  %8.3f ms  %6.2f\n", d5, d5 * 1e-3 * 2.4e9 / (6.0 * pairs_per_wave));
    printf("LDS records + DPP quad broadcast    %8.3f ms  %6.2f\n", d6, d6 * 1e-3 * 2.4e9 / (6.0 * pairs_per_wave));
    printf("mfma 32x32x2 + 2 VALU, 4 waves/SIMD %8.3f ms  %6.2f\n", c, c * 1e-3 * 2.4e9 / (4.0 * pairs_per_wave));
    return 0;
}
