// The plane filter of the render kernel in isolation, with the dot product of every (sphere, ray) pair on the bf16 matrix pipe,
// at the render kernel's occupancy (512-thread blocks, 3 per CU = 6 waves per SIMD):
//   mode 0  the scalar-operand form (two FMAs with an SGPR operand per pair, then verdict and accumulate): the vector-ALU form (RT_AMD_FILTER=plane);
//   mode 1  v_mfma_f32_32x32x16_bf16: q[sphere i][ray j] = sum over 16 bf16 terms (the f32 factors split into bf16 pieces), for
//           32 spheres x 32 rays per instruction, two instructions per chunk of 32 spheres (rays 0-31, rays 32-63).  A operand
//           (spheres) 16 bytes per lane per chunk from LDS, B operands (rays) built once per sweep with v_permlane32_swap.  Lane l
//           of instruction 1 ends up with ray l % 32 against spheres 16 (l / 32) .. + 15; verdict = clamp(T - q q), acc = 2 acc +
//           verdict on the vector ALU (all operands vector registers); one v_permlane32_swap per chunk gives every lane the two
//           16-bit halves of ITS ray's mask word, which goes to LDS like the shipped filter's;
//   mode 2  as mode 1 with the A operand read from global memory (L1 / L2) instead of LDS, one chunk ahead;
//   mode 3  v_mfma_f32_32x32x8_bf16_1k (K = 8, 8 bytes per lane for A), otherwise as mode 1;
//   modes 4, 5  the MFMAs of mode 1 alone / the vector-ALU work of mode 1 alone; mode 6  mode 1 with the next half's MFMA issued early;
//   mode 7  the form the render kernel ships: mode 3 with the accumulate chain started at 128 (no conversions) and v_perm_b32.
// Reported: SIMD cycles per wave-pair (one sphere against the 64 rays of a wave) = time x clock x SIMDs / (waves x pairs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s4v __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) f4v* Rec4;

__device__ __forceinline__ float verdict_of(float q, float T)
{
    // the compiler folds the [0, 1] clamp into the FMA's clamp modifier and keeps the wait states the read of an MFMA result needs
    return __builtin_amdgcn_fmed3f(fmaf(-q, q, T), 0.0f, 1.0f);
}
__device__ __forceinline__ void swap32(uint32_t& a, uint32_t& b)      // a's lanes 32-63 <-> b's lanes 0-31
{
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0]; b = r[1];
}
// x = p0 + p1 + p2 with every piece a bf16 number (top 16 bits of a float), by truncation: exact, |x - sum| < 2^-21 |x|
__device__ __forceinline__ void split3(float x, uint32_t& p0, uint32_t& p1, uint32_t& p2)
{
    p0 = __float_as_uint(x) & 0xffff0000u;
    const float r1 = x - __uint_as_float(p0);
    p1 = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(p1);
    p2 = __float_as_uint(r2) & 0xffff0000u;
}

template <int MODE, int WAVES>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) k(const float* recs, const uint4* aimg, uint32_t* out, int chunks, int trips)
{
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* words = lds + wave * 16 * 64 + lane;
    uint4* lds_a = reinterpret_cast<uint4*>(lds + 8 * 16 * 64);
    float ux = (0.6f + threadIdx.x * 1e-4f) * 0x1p44f, uz = (0.8f - threadIdx.x * 1e-4f) * 0x1p44f, nou = -0.3f * 0x1p44f;
    float T = 0x1p88f;
    asm volatile("" : "+v"(T));
    uint32_t sum = 0;
    if (MODE != 0 && MODE != 2) { for (int i = threadIdx.x; i < chunks * 64; i += 512) lds_a[i] = aimg[i]; __syncthreads(); }
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
        } else {
            // the ray's sixteen terms: k 0-7 in L, k 8-15 in H (two bf16 per register)
            uint32_t x0, x1, x2, z0, z1, z2, n0, n1, n2;
            split3(ux, x0, x1, x2); split3(uz, z0, z1, z2); split3(nou, n0, n1, n2);
            // pairing with the sphere side (c0 c0 c0 c1 c1 c2 | same for z | r r r 0): x0 x1 x2 x0 x1 x0 ...
            uint32_t L[4], H[4];
            L[0] = (x0 >> 16) | x1; L[1] = (x2 >> 16) | x0; L[2] = (x1 >> 16) | x0; L[3] = (z0 >> 16) | z1;
            H[0] = (z2 >> 16) | z0; H[1] = (z1 >> 16) | z0; H[2] = (n0 >> 16) | n1; H[3] = (n2 >> 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) swap32(L[i], H[i]);          // L: operand for rays 0-31, H: operand for rays 32-63
            f16v Z;
#pragma unroll
            for (int v = 0; v < 16; ++v) Z[v] = 0.0f;
            uint32_t nz = 0;
            if (MODE == 3) {
                s4v b1, b2;
                b1[0] = (short)L[0]; b1[1] = (short)(L[0] >> 16); b1[2] = (short)L[1]; b1[3] = (short)(L[1] >> 16);
                b2[0] = (short)H[0]; b2[1] = (short)(H[0] >> 16); b2[2] = (short)H[1]; b2[3] = (short)(H[1] >> 16);
                const uint2* lds_a2 = reinterpret_cast<const uint2*>(lds_a);
                for (int c = 0; c < chunks; ++c) {
                    const uint2 araw = lds_a2[c * 64 + lane];
                    s4v a; a[0] = (short)araw.x; a[1] = (short)(araw.x >> 16); a[2] = (short)araw.y; a[3] = (short)(araw.y >> 16);
                    const f16v d1 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b1, Z, 0, 0, 0);
                    const f16v d2 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b2, Z, 0, 0, 0);
                    float acc1 = 0.0f, acc2 = 0.0f;
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc1 = fmaf(acc1, 2.0f, verdict_of(d1[v], T));
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc2 = fmaf(acc2, 2.0f, verdict_of(d2[v], T));
                    uint32_t m1 = (uint32_t)acc1, m2 = (uint32_t)acc2;
                    swap32(m1, m2);
                    const uint32_t m = (m1 << 16) | m2;
                    words[c * 64] = m; nz |= m != 0u ? 0x80000000u >> c : 0u;
                }
            } else if (MODE == 7) {
                // the shipped form: K = 8 instruction, chain started at 128 (verdict bits end up in the float's low half: no conversion),
                // one v_perm_b32 to join the halves, survivor flag by add-with-carry
                s4v b1, b2;
                b1[0] = (short)L[0]; b1[1] = (short)(L[0] >> 16); b1[2] = (short)L[1]; b1[3] = (short)(L[1] >> 16);
                b2[0] = (short)H[0]; b2[1] = (short)(H[0] >> 16); b2[2] = (short)H[1]; b2[3] = (short)(H[1] >> 16);
                const uint2* lds_a2 = reinterpret_cast<const uint2*>(lds_a);
                uint2 anext = lds_a2[lane];
                uint32_t seen = 0;
                for (int c = 0; c < chunks; ++c) {
                    const uint2 araw = anext;
                    anext = lds_a2[(c + 1) * 64 + lane];
                    s4v a; a[0] = (short)araw.x; a[1] = (short)(araw.x >> 16); a[2] = (short)araw.y; a[3] = (short)(araw.y >> 16);
                    const f16v d1 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b1, Z, 0, 0, 0);
                    const f16v d2 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b2, Z, 0, 0, 0);
                    float acc1 = 128.0f, acc2 = 128.0f;
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc1 = fmaf(acc1, 2.0f, verdict_of(d1[v], T));
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc2 = fmaf(acc2, 2.0f, verdict_of(d2[v], T));
                    uint32_t m1 = __float_as_uint(acc1), m2 = __float_as_uint(acc2);
                    swap32(m1, m2);
                    const uint32_t m = __builtin_amdgcn_perm(m1, m2, 0x05040100u);
                    words[c * 64] = m; seen = seen + seen + (m != 0u ? 1u : 0u);
                }
                nz = seen;
            } else if (MODE == 4) {
                // the MFMAs alone, accumulating in place
                bf16x8 b1, b2;
                { uint4 u = make_uint4(L[0], L[1], L[2], L[3]); b1 = __builtin_bit_cast(bf16x8, u); u = make_uint4(H[0], H[1], H[2], H[3]); b2 = __builtin_bit_cast(bf16x8, u); }
                f16v d1 = Z, d2 = Z;
                for (int c = 0; c < chunks; ++c) {
                    const bf16x8 a = __builtin_bit_cast(bf16x8, lds_a[c * 64 + lane]);
                    d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, d1, 0, 0, 0);
                    d2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b2, d2, 0, 0, 0);
                }
#pragma unroll
                for (int v = 0; v < 16; ++v) nz ^= __float_as_uint(d1[v]) ^ __float_as_uint(d2[v]);
            } else if (MODE == 5) {
                // the vector-ALU work alone, on registers no MFMA writes
                f16v d1, d2;
#pragma unroll
                for (int v = 0; v < 16; ++v) { d1[v] = __uint_as_float(L[v & 3] + v); d2[v] = __uint_as_float(H[v & 3] + v); }
                for (int c = 0; c < chunks; ++c) {
                    float acc1 = 0.0f, acc2 = 0.0f;
                    asm volatile("" : "+v"(d1), "+v"(d2));
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc1 = fmaf(acc1, 2.0f, verdict_of(d1[v], T));
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc2 = fmaf(acc2, 2.0f, verdict_of(d2[v], T));
                    uint32_t m1 = (uint32_t)acc1, m2 = (uint32_t)acc2;
                    swap32(m1, m2);
                    const uint32_t m = (m1 << 16) | m2;
                    words[c * 64] = m; nz |= m != 0u ? 0x80000000u >> c : 0u;
                }
            } else if (MODE == 6) {
                // one MFMA always in flight: the next half's instruction is issued before the current half's results are worked on
                bf16x8 b1, b2;
                { uint4 u = make_uint4(L[0], L[1], L[2], L[3]); b1 = __builtin_bit_cast(bf16x8, u); u = make_uint4(H[0], H[1], H[2], H[3]); b2 = __builtin_bit_cast(bf16x8, u); }
                bf16x8 a = __builtin_bit_cast(bf16x8, lds_a[lane]);
                f16v d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, Z, 0, 0, 0);
                for (int c = 0; c < chunks; ++c) {
                    const f16v d2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b2, Z, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    a = __builtin_bit_cast(bf16x8, lds_a[(c + 1) * 64 + lane]);
                    float acc1 = 0.0f, acc2 = 0.0f;
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc1 = fmaf(acc1, 2.0f, verdict_of(d1[v], T));
                    __builtin_amdgcn_sched_barrier(0);
                    d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, Z, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc2 = fmaf(acc2, 2.0f, verdict_of(d2[v], T));
                    uint32_t m1 = (uint32_t)acc1, m2 = (uint32_t)acc2;
                    swap32(m1, m2);
                    const uint32_t m = (m1 << 16) | m2;
                    words[c * 64] = m; nz |= m != 0u ? 0x80000000u >> c : 0u;
                    __builtin_amdgcn_sched_barrier(0);
                }
                nz ^= __float_as_uint(d1[0]);
            } else {
                bf16x8 b1, b2;
                { uint4 u = make_uint4(L[0], L[1], L[2], L[3]); b1 = __builtin_bit_cast(bf16x8, u); u = make_uint4(H[0], H[1], H[2], H[3]); b2 = __builtin_bit_cast(bf16x8, u); }
                uint4 anext = MODE == 2 ? aimg[lane] : make_uint4(0, 0, 0, 0);
                for (int c = 0; c < chunks; ++c) {
                    uint4 araw;
                    if (MODE == 2) { araw = anext; anext = aimg[(c + 1) * 64 + lane]; }
                    else araw = lds_a[c * 64 + lane];
                    const bf16x8 a = __builtin_bit_cast(bf16x8, araw);
                    const f16v d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, Z, 0, 0, 0);
                    const f16v d2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b2, Z, 0, 0, 0);
                    float acc1 = 0.0f, acc2 = 0.0f;
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc1 = fmaf(acc1, 2.0f, verdict_of(d1[v], T));
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc2 = fmaf(acc2, 2.0f, verdict_of(d2[v], T));
                    uint32_t m1 = (uint32_t)acc1, m2 = (uint32_t)acc2;
                    swap32(m1, m2);
                    const uint32_t m = (m1 << 16) | m2;
                    words[c * 64] = m; nz |= m != 0u ? 0x80000000u >> c : 0u;
                }
            }
            sum ^= nz;
            for (int c = 0; c < chunks; c += 5) sum ^= words[c * 64];
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = sum;
}

template <int MODE, int WAVES> float run(const float* recs, const uint4* aimg, uint32_t* out, int chunks, int trips, hipEvent_t e0, hipEvent_t e1)
{
    float best = 1e30f;
    const int blocks_per_cu = WAVES / 2;
    const size_t lds = WAVES == 6 ? 52248 : WAVES == 4 ? 65536 : 40000;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, WAVES>), dim3(256 * blocks_per_cu), dim3(512), lds, 0, recs, aimg, out, chunks, trips);
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
    // A image: random bf16 pairs of moderate size
    std::vector<uint32_t> ha((chunks + 2) * 64 * 4);
    for (size_t i = 0; i < ha.size(); ++i) { const uint32_t r = (uint32_t)(i * 2654435761u); ha[i] = (0x3f80u + (r & 0x7fu)) | ((0x3f00u + ((r >> 8) & 0xffu)) << 16) | (r & 0x80008000u); }
    uint4* aimg; CHECK(hipMalloc(&aimg, ha.size() * 4)); CHECK(hipMemcpy(aimg, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
    uint32_t* out; CHECK(hipMalloc(&out, 256 * 4 * 512 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int trips = 2000;
    const double pairs_per_wave = (double)trips * chunks * 32;
    auto cyc = [&](float ms, double waves) { return ms * 1e-3 * 2.4e9 / (waves * pairs_per_wave); };
    const float a = run<0, 6>(recs, aimg, out, chunks, trips, e0, e1);
    printf("scalar-operand form, 6 waves/SIMD            %8.3f ms  %6.2f SIMD cycles per wave-pair (at 2.4 GHz)\n", a, cyc(a, 6));
    const float b = run<1, 6>(recs, aimg, out, chunks, trips, e0, e1);
    printf("mfma 32x32x16 bf16, A from LDS, 6 waves/SIMD %8.3f ms  %6.2f\n", b, cyc(b, 6));
    const float c = run<2, 6>(recs, aimg, out, chunks, trips, e0, e1);
    printf("mfma 32x32x16 bf16, A from L1,  6 waves/SIMD %8.3f ms  %6.2f\n", c, cyc(c, 6));
    const float d = run<3, 6>(recs, aimg, out, chunks, trips, e0, e1);
    printf("mfma 32x32x8 bf16_1k, A from LDS, 6 w/SIMD   %8.3f ms  %6.2f\n", d, cyc(d, 6));
    const float e = run<1, 4>(recs, aimg, out, chunks, trips, e0, e1);
    printf("mfma 32x32x16 bf16, A from LDS, 4 waves/SIMD %8.3f ms  %6.2f\n", e, cyc(e, 4));
    const float f = run<4, 6>(recs, aimg, out, chunks, trips, e0, e1);
    printf("the MFMAs alone (32x32x16 bf16), 6 w/SIMD      %8.3f ms  %6.2f\n", f, cyc(f, 6));
    const float g = run<5, 6>(recs, aimg, out, chunks, trips, e0, e1);
    printf("the vector-ALU work alone, 6 waves/SIMD      %8.3f ms  %6.2f\n", g, cyc(g, 6));
    const float h6 = run<6, 6>(recs, aimg, out, chunks, trips, e0, e1);
    printf("mfma 32x32x16, next half issued early, 6 w/S %8.3f ms  %6.2f\n", h6, cyc(h6, 6));
    const float s7 = run<7, 6>(recs, aimg, out, chunks, trips, e0, e1);
    printf("shipped: 32x32x8 bf16_1k, no conversions     %8.3f ms  %6.2f\n", s7, cyc(s7, 6));
    CHECK(hipGetLastError());
    return 0;
}
