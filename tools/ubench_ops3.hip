// Per-instruction issue cost on gfx950 at 8 waves/SIMD, third set: the integer multiplies of Philox4x32-10 (one 64-bit
// product against a low/high pair, 24-bit multiplies), and whole Philox blocks written both ways.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void mul64(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo)
{
    uint64_t p, carry;
    asm volatile("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p), "=s"(carry) : "v"(a), "v"(b));
    lo = (uint32_t)p; hi = (uint32_t)(p >> 32);
}

template <int MODE>
__global__ void k(uint32_t* out, int iters, uint32_t bb, uint32_t cc)
{
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    uint32_t b = bb + threadIdx.x, c = cc;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) {        // Philox round pair, separate low and high multiplies
                const uint32_t h0 = __umulhi(0xD2511F53u, a0), l0 = 0xD2511F53u * a0;
                const uint32_t h1 = __umulhi(0xCD9E8D57u, a2), l1 = 0xCD9E8D57u * a2;
                const uint32_t n0 = h1 ^ a1 ^ b, n2 = h0 ^ a3 ^ c;
                a0 = n0; a1 = l1; a2 = n2; a3 = l0; b += 0x9E3779B9u; c += 0xBB67AE85u;
            }
            if (MODE == 1) {        // the same round with one 64-bit product per multiplier
                uint32_t h0, l0, h1, l1;
                mul64(a0, 0xD2511F53u, h0, l0);
                mul64(a2, 0xCD9E8D57u, h1, l1);
                const uint32_t n0 = h1 ^ a1 ^ b, n2 = h0 ^ a3 ^ c;
                a0 = n0; a1 = l1; a2 = n2; a3 = l0; b += 0x9E3779B9u; c += 0xBB67AE85u;
            }
            if (MODE == 2) { asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a0) : "v"(b)); asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a1) : "v"(b));
                             asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a2) : "v"(b)); asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a3) : "v"(b)); }
            if (MODE == 3) { asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a0) : "v"(b)); asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a1) : "v"(b));
                             asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a2) : "v"(b)); asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a3) : "v"(b)); }
            if (MODE == 4) { asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a0) : "v"(b)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a1) : "v"(b));
                             asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a2) : "v"(b)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a3) : "v"(b)); }
            if (MODE == 5) { uint32_t h, l; mul64(a0, b, h, l); a0 = l; a1 ^= h; mul64(a2, b, h, l); a2 = l; a3 ^= h;
                             mul64(a0, c, h, l); a0 = l; a1 ^= h; mul64(a2, c, h, l); a2 = l; a3 ^= h; }
            if (MODE == 6) { asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c)); asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c));
                             asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c)); asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c)); }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ b ^ c;
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
    const char* names[7] = { "philox round (mul_lo + mul_hi)", "philox round (v_mad_u64_u32)", "v_mul_u32_u24", "v_mul_hi_u32_u24", "v_mul_lo_u32", "v_mad_u64_u32 (+1 xor each)", "v_mad_u32_u24" };
    const double per_iter[7] = { 8, 8, 32, 32, 32, 32, 32 };     // rounds or instructions per loop iteration and lane
    float ms[7];
    ms[0] = run<0>(out, iters, blocks, e0, e1); ms[1] = run<1>(out, iters, blocks, e0, e1); ms[2] = run<2>(out, iters, blocks, e0, e1);
    ms[3] = run<3>(out, iters, blocks, e0, e1); ms[4] = run<4>(out, iters, blocks, e0, e1); ms[5] = run<5>(out, iters, blocks, e0, e1);
    ms[6] = run<6>(out, iters, blocks, e0, e1);
    // 2048 blocks x 4 waves on 1024 SIMDs = 8 waves per SIMD; cycles at 2.4 GHz per wave-level unit
    for (int m = 0; m < 7; ++m) printf("%-34s %8.3f ms  %6.2f SIMD cycles per unit\n", names[m], ms[m], ms[m] * 1e-3 * 2.4e9 / (8.0 * iters * per_iter[m]));
    return 0;
}
