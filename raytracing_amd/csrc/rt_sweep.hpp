// Device code shared by the render kernels of librt_amd.so (rt_pixel.hip, rt_trace.hip, rt_wavefront.hip): camera rays,
// the closest-hit sweep in its three forms (all lanes busy, few lanes left, hierarchy walk), one bounce of Color(), and
// the numbering of work items.  Everything is forceinline; every translation unit gets its own copy.
#pragma once
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdint>
#include "rt_internal.hpp"
#include "rt_bvh.hpp"

// ============================================================================ device data layout
// Scene records in HBM (built once per upload) and, for the megakernel, mirrored into LDS:
//   geom[i] = (cx, cy, cz, r*r)       16 B  -- everything the sweep touches
//   mat[i]  = (ar, ag, ab, param)     16 B  -- read once per hit
//   aux[i]  = (radius, type)           8 B  -- read once per hit (normal = (p - c) / radius)


static constexpr int kBlock = 256;          // 4 waves; each wave owns an 8x8 pixel tile
static constexpr int kLdsBytesPerSphere = 40;
static constexpr int kMaxLdsSpheres = 1536; // 60 KB of scene records per block; with the 32 KB of mask words of a 512-thread block (92 KB) one block per CU at this size, three at the 487 spheres of the final scene

// Compact owned row r -> image row y (from the top).
__device__ __forceinline__ int owned_row_to_y2(int band_rows, int band_first, int band_step, int r)
{
    const int k = r / band_rows;
    return (band_first + k * band_step) * band_rows + (r - k * band_rows);
}

// Camera ray of (pixel, sample): jitter from words 0,1 of group 0 (main.cpp:126-127), lens sample by
// rejection from the words lens_words() names, then Camera::GetRay (camera.cpp:46-53).
template <bool kWide = false>
__device__ __forceinline__ RayD camera_ray(const CameraD& cam, uint32_t pix, uint32_t smp, int i, int j, float fnx, float fny, uint32_t k0, uint32_t k1)
{
    U4 blk = philox4x32_10_t<kWide>(pix, smp, 0u, 0u, k0, k1);
    const float fu = (float)((double)i + (double)u01(blk.x)) / fnx;
    const float fv = (float)((double)j + (double)u01(blk.y)) / fny;
    float dx = 0.0f, dy = 0.0f;
    if (cam.lens_radius > 0.0f) {
        uint32_t t = 0;
        for (;;) {
            float a, b;
            lens_words(blk, t, a, b);
            dx = 2.0f * a - 1.0f; dy = 2.0f * b - 1.0f;
            if (!((dx * dx) + (dy * dy) + 0.0f >= 1.0f)) break;
            ++t;
            if (t & 1u) blk = philox4x32_10_t<kWide>(pix, smp, 0u, (t + 1u) >> 1, k0, k1);
        }
    }
    return get_ray(cam, fu, fv, dx, dy);
}

// Device view of the sphere hierarchy (rt_bvh.hpp) for RT_MODE_BVH.
struct BvhScene {
    const float4* nodes; const uint32_t* leaf_index; const uint32_t* big; int n_big;
    const float4* geom;
};

// Diagnostic build (-DRT_SECTION_TIMERS): every wave accumulates s_memtime ticks per section of the
// main loop into g_section_ticks.  Never enabled in the shipped library (mark() is empty there).
#ifdef RT_WAVE_TIMELINE
// Diagnostic build (-DRT_WAVE_TIMELINE), per wave of the last k_trace launch: s_memrealtime (100 MHz, common to all XCDs) at entry, when the wave first found the work counter dry, at exit;
// and how many trips of the main loop it made after that point
static __device__ unsigned long long g_wave_times[10][8192];     // rows 4, 5: trips made with the sparse sweep, ticks spent in those trips (whole trip); 6: first sweep starts; 7: first trip ends; 8: the block's shared end begins (0: never); 9: its bounces
#endif
#ifdef RT_SECTION_TIMERS
static constexpr int kSections = 12;
static __device__ unsigned long long g_section_ticks[kSections];
struct SecTimer {
    unsigned long long acc[kSections]; unsigned long long last;
    __device__ __forceinline__ void start() { for (int k = 0; k < kSections; ++k) acc[k] = 0; last = __builtin_amdgcn_s_memtime(); }
    __device__ __forceinline__ void mark(int idx) { const unsigned long long t = __builtin_amdgcn_s_memtime(); acc[idx] += t - last; last = t; }
    __device__ __forceinline__ void flush(int lane) { if (lane == 0) for (int k = 0; k < kSections; ++k) atomicAdd(&g_section_ticks[k], acc[k]); }
};
#else
struct SecTimer {
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void flush(int) {}
};
#endif
#ifndef RT_CHUNK_ITEMS
#define RT_CHUNK_ITEMS 512
#endif
static constexpr int kChunkItems = RT_CHUNK_ITEMS;
static constexpr int kStagedFloats = 12;       // one staged ray: 48 bytes (origin, direction, pixel key, sample, sample-buffer slot, first sphere, 8 bytes of padding), written whole (three 16-byte stores) so that no line goes back to HBM partly written
static constexpr int kTraceBlock = 512;         // 8 waves share one LDS copy of the scene records
// k_trace with the matrix-pipe filter keeps that filter's A operands in LDS too (8 KB more at 487 spheres): 12 waves share one copy,
// two such blocks per CU are the same six waves per SIMD
#ifndef RT_MFMA_BLOCK
#define RT_MFMA_BLOCK 768
#endif
__host__ __device__ constexpr int trace_block(bool mfma) { return mfma ? RT_MFMA_BLOCK : kTraceBlock; }
// A staged ray is read back by another lane of the wave that wrote it, and its slot is rewritten for every chunk: the loads are
// device-scope ones (sc1), so that a line of the previous chunk still in this CU's L1 is not served.  Three loads off one address
// and one wait for the lot.
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void staged_read(const float* rec, v4f& a, v4f& b, v2f& c)
{
    asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %3, off offset:16 sc1\n\tglobal_load_dwordx2 %2, %3, off offset:32 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c) : "v"(rec) : "memory");
}
#ifndef RT_SUPER_WORDS
#define RT_SUPER_WORDS 16
#endif
static constexpr int kSuperChunkWords = RT_SUPER_WORDS;         // 32-sphere mask words kept per lane between exact passes (16 words = 512 spheres); 16, 8 or 4
static constexpr float kRaySlack = 160.0f * 5.9604645e-08f;   // per-ray slack, relative to |o|^2
static constexpr float kPlaneRaySlack = 144.0f * 5.9604645e-08f;
static constexpr float kPlaneScale = 0x1p44f;                   // scaling of the plane filter's per-ray constants; its records carry 2^88 R

// ============================================================================ closest-hit sweep
// HittableList::Hit (hittablelist.cpp:13-31) over Sphere::Hit (sphere.cpp:15-52) for one ray per lane,
// brute force over all spheres, as filter + exact (see k_trace).  Shared by the persistent megakernel and
// the wavefront intersect kernel.  mask_words: this lane's column of kSuperChunkWords LDS words.
struct SweepScene {
    const float4* frec;      // disc-filter records (cx, cy, cz, |c|^2 - r^2 - slack), padded to a multiple of 8
    const float4* prec;      // plane-filter records (cx, cz, R_i = r^2 + slack_i, 0), same padding
    float plane_s0;          // constant part of the plane filter's per-ray slack
    float plane_rmin;        // smallest R_i over the non-degenerate spheres
    const uint32_t* force;   // per 32 spheres: bits (31 = first) of the spheres the plane filter does not judge (always candidates)
    int n_forced;            // how many bits are set in `force`
    const float4* geom;      // exact records in HBM (used when they are not mirrored in LDS)
    int n, n_pad;
    const uint2* arec;       // matrix-pipe form of the plane filter: per chunk of 32 spheres, 64 x 8 bytes (one MFMA A operand, see filter_mfma)
    float mfma_g;            // per-ray allowance of that form: kappa is scaled by 1 - (|o.x| + |o.z|) mfma_g
};

// LDS mirror of the records the exact passes read.  Packed form (k_trace): geom[k] = (centre, constant of the discriminant
// filter) and r*r is formed from aux[k].radius -- the same single float product the host stores in the HBM record -- so a sphere
// costs no more LDS than before the filter constant moved in.  Plain form (wavefront intersect): geom[k] = (centre, r*r) as in
// HBM and the constants in an array of their own.
struct ExactLds { const float4* geom; const Aux* aux; const float* disck; const uint2* arec = nullptr; /* LDS copy of SweepScene::arec (k_trace, matrix-pipe filter) */ };
template <bool kFromRadius>
__device__ __forceinline__ float exact_r2(const float4& g, const ExactLds& lds, int k)
{
    if (kFromRadius) { const float r = lds.aux[k].radius; return r * r; }
    return g.w;
}

// s_setprio takes an immediate
__device__ __forceinline__ void set_wave_priority(int p)
{
    if (p >= 3) __builtin_amdgcn_s_setprio(3);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else if (p == 1) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}
typedef float f16acc __attribute__((ext_vector_type(16)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
// a's lanes 32-63 <-> b's lanes 0-31
__device__ __forceinline__ void swap_halves(uint32_t& a, uint32_t& b)
{
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0]; b = r[1];
}
// x rounded to the nearest bf16 (ties to even), as the bits of a float (low half zero).  Not for NaN.
__device__ __forceinline__ uint32_t bf16_bits(float x)
{
    const uint32_t b = __float_as_uint(x);
    return (b + 0x7fffu + ((b >> 16) & 1u)) & 0xffff0000u;
}

#ifndef RT_PASSA_ADDR
#define RT_PASSA_ADDR 1
#endif
// Large scenes, LDS-tiled (kTiled; k_trace only): the matrix-pipe filter's operands of one super-chunk -- 512 spheres, 8 KB -- are a
// TILE that the block stages in LDS once for its twelve waves, instead of every wave streaming all of them from L2 for itself.
// Two tile buffers; the waves of a block sweep in step: at the top of every super-chunk they wait for the tile that was requested
// one super-chunk earlier (global_load_lds: memory to LDS without passing through registers), meet at a barrier, and request the
// next one into the other buffer -- cyclically, the first tile of the next sweep behind the last of this one, so that the
// sequence of buffers never breaks (`tiles_done` counts tiles over the whole launch; tile g lies in buffer g & 1).  The caller
// has done wait, barrier and count for the first super-chunk of the sweep.  A tile is requested by the block's first eight waves,
// 16 bytes per lane; chunks beyond the scene's last are not requested (and never read).
constexpr int kTileSpheres = 32 * 16;                 // one super-chunk
constexpr int kTileEntries = kTileSpheres / 32 * 64;  // uint2 entries per tile buffer (8 KB)
__device__ __forceinline__ void stage_operand_tile(const SweepScene& sw, const uint2* tile_buffers, int tile, uint32_t buffer)
{
    typedef __attribute__((address_space(1))) const void* GlobalPtr;
    typedef __attribute__((address_space(3))) void* LdsPtr;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (wave < kTileEntries * 8 / (64 * 16)) {
        const int lane = threadIdx.x & 63;
        const int chunk = tile * (kTileSpheres / 32) + wave * 2 + (lane >> 5);          // 32 lanes x 16 bytes = one chunk's 512 bytes
        if (chunk < ((sw.n + 31) >> 5))
            __builtin_amdgcn_global_load_lds((GlobalPtr)(reinterpret_cast<const char*>(sw.arec) + (size_t)tile * (kTileEntries * 8) + (size_t)(wave * 64 + lane) * 16),
                                             (LdsPtr)(reinterpret_cast<const char*>(tile_buffers) + (size_t)buffer * (kTileEntries * 8) + (size_t)wave * 1024), 16, 0, 0);
    }
}
template <bool kLds, bool kPlane, bool kPacked, bool kMfma = false, bool kTiled = false>
__device__ __forceinline__ void closest_hit(const RayD& ray, float tmin, const SweepScene& sw, const ExactLds& lds,
                                            uint32_t* mask_words, float& closest, int& hit, SecTimer& T, bool active = true, int prio_after = 0,
                                            uint32_t* tiles_done = nullptr)
{
    static_assert(!kTiled || (kMfma && !kLds), "operand tiles are for the matrix-pipe filter of scenes whose records do not fit LDS");
    // prio_after (wave-uniform, matrix-pipe form only): the filter loop is a long run of independent instructions that can fill any
    // issue slot, the exact passes and the shading that follow are chains of dependent ones; the wave drops to issue priority 0 for
    // the filter and returns to prio_after behind it, so that on a SIMD the waves in their latency-bound sections issue first and
    // the filtering waves take what is left (config 3: 80.1 -> 76.9 ms).
    // kMfma: the plane filter's dot product runs on the matrix pipe (filter_mfma below).  The matrix instruction and the
    // half-wave swaps around it work on the whole wave, so in this form the function is entered by ALL 64 lanes of the wave;
    // a lane without a ray (active == false) is given a NaN normal, which fails every test, and leaves with hit = -1.
    static_assert(!kMfma || kPlane, "the matrix-pipe filter is a form of the plane filter");
    // Per ray: unit direction (approximate is enough for a conservative test), -(o.d^), 2o,
    // -|o|^2 (1 - slack).  Per sphere record (cx, cy, cz, k), k = |c|^2 - r^2 - slack_i:
    //   b = c.d^ - o.d^            3 FMA      w = 2 o.c - |o|^2'        3 FMA
    //   e = b*b + w                1 FMA      s = e - k                 1 SUB   -> sign bit
    // s >= 0 is necessary for the reference's discriminant to be positive.
    const float qa = dot(ray.d, ray.d);
    const float rs = __frsqrt_rn(fmaf(ray.d.x, ray.d.x, fmaf(ray.d.y, ray.d.y, ray.d.z * ray.d.z)));
    const float hx = ray.d.x * rs, hy = ray.d.y * rs, hz = ray.d.z * rs;
    const float nod = -fmaf(ray.o.x, hx, fmaf(ray.o.y, hy, ray.o.z * hz));
    const float p2x = 2.0f * ray.o.x, p2y = 2.0f * ray.o.y, p2z = 2.0f * ray.o.z;
    const float oo = fmaf(ray.o.x, ray.o.x, fmaf(ray.o.y, ray.o.y, ray.o.z * ray.o.z));
    const float noo = kRaySlack * oo - oo;
    // Plane filter (kPlane): the vertical plane through the ray has a horizontal normal u perp d, so
    // the signed distance of a centre to it is pu = cx*ux + cz*uz - o.u (2 FMA).  The reference
    // accepts a sphere only if its discriminant b^2 - a c is positive, and in its float arithmetic
    // that can happen for a line passing up to sqrt(r^2 + 32 ulp (|o-c|^2 + r^2)) from the centre,
    // so the test is pu^2 <= R_i + S_ray with that allowance (and this form's own rounding) split
    // into a per-sphere part R_i (host) and a per-ray part S_ray.
    float ux = 1.0f, uz = 0.0f, nou = 0.0f, s_ray = 0.0f;
    if (kPlane) {
        const float h2 = fmaf(hx, hx, hz * hz);
        if (h2 > 1e-12f) { const float ih = __frsqrt_rn(h2); ux = hz * ih; uz = -hx * ih; }
        // Fold the per-ray allowance S_ray into the normal instead of subtracting it per pair: with
        // kappa^2 = Rmin / (Rmin + S_ray), (kappa pu)^2 <= R_i holds whenever pu^2 <= R_i + S_ray and
        // R_i >= Rmin, so the per-pair test is the sign of R_i - pu'^2 alone (3 FMA per pair).
        s_ray = fmaf(kPlaneRaySlack, oo, sw.plane_s0);
        const float k2 = sw.plane_rmin / (sw.plane_rmin + s_ray);
        float kappa = sqrtf(k2) * (1.0f - 4.0f * 5.9604645e-08f);
        // The bf16 pieces the matrix pipe multiplies leave pu off by up to 2^-13 (|cx| + |cz| + |o.u|): the sphere's part is in its
        // record (host), the ray's part shrinks the normal once more (|o.u| <= |o.x| + |o.z|; kappa = 0 passes every sphere)
        if (kMfma) kappa *= fmaxf(0.0f, 1.0f - (fabsf(ray.o.x) + fabsf(ray.o.z)) * sw.mfma_g);
        ux *= kappa; uz *= kappa;
        nou = -fmaf(ray.o.x, ux, ray.o.z * uz);
        // The verdict of a pair is wanted as the number 0.0 or 1.0 (see filter_bits): everything is scaled by 2^44 (exact),
        // which makes R' - pu'^2 = 2^88 (R - pu^2) either <= 0 or >= 1 for every sphere the host admits to this filter.
        ux *= kPlaneScale; uz *= kPlaneScale; nou *= kPlaneScale;
    }
    // Filter records are read through the constant address space: the index is wave-uniform,
    // so each record arrives by scalar load in SGPRs and feeds the FMAs as a scalar operand.
    typedef float f4v __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) f4v* ConstRec;
    const ConstRec frec_k = (ConstRec)(sw.frec);
    const ConstRec prec_k = (ConstRec)(sw.prec);

    // Filter one chunk of kCount spheres (32, or 8 / 16 / 24 at the end of the list): candidate bits, left-aligned (bit 31 =
    // first sphere of the chunk), 1 = may be hit.
    //
    // Plane filter: s = R - pu pu, pu = cx ux + (cz uz - o.u), candidate when s > 0.  A sign bit would have to be moved
    // into a mask by an integer instruction (v_alignbit_b32, four cycles of the vector pipe against two for an FMA).
    // Instead the third FMA carries the clamp modifier, which is free: with the 2^88 scaling its result is exactly 1.0
    // for a candidate and 0.0 otherwise (-inf and NaN clamp to 0), and a fourth FMA, acc = 2 acc + s, shifts it into a
    // float that holds up to sixteen verdicts as an integer.  Why no value falls strictly between 0 and 1: R - pu pu is
    // formed exactly inside the FMA and is a multiple of min(ulp(R), ulp(pu)^2); it can only be small when pu^2 is
    // close to R, and for 2^-38 <= R <= 2^38 that puts both granules at 2^-88 or more, so a positive difference is at
    // least 2^-88, i.e. at least 1 after scaling.  Spheres outside that range of R (and zero-radius ones) are not
    // judged here at all: the host gives them a record that never passes and sets their bit in `force`.
    // s = 0 exactly (pu^2 equal to the inflated threshold) counts as a miss, which the margin inside R allows.
    auto filter_bits = [&](int base, auto count_tag) -> uint32_t {
        constexpr int kCount = decltype(count_tag)::value;
        if (kPlane) {
            float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll
            for (int jj = 0; jj < kCount; ++jj) {
                const f4v f = prec_k[base + jj];
                const float pu = fmaf(f.x, ux, fmaf(f.y, uz, nou));
                float verdict;
                asm("v_fma_f32 %0, -%1, %1, %2 clamp" : "=v"(verdict) : "v"(pu), "s"(f.z));
                if (jj < 16) acc0 = fmaf(acc0, 2.0f, verdict); else acc1 = fmaf(acc1, 2.0f, verdict);
            }
            const uint32_t bits = kCount > 16 ? ((uint32_t)acc0 << (kCount > 16 ? kCount - 16 : 0)) | (uint32_t)acc1 : (uint32_t)acc0;
            return bits << (32 - kCount);
        }
        uint32_t mask = 0u;
#pragma unroll
        for (int jj = 0; jj < kCount; ++jj) {
            const f4v f = frec_k[base + jj];
            const float b = fmaf(f.x, hx, fmaf(f.y, hy, fmaf(f.z, hz, nod)));
            const float w = fmaf(f.x, p2x, fmaf(f.y, p2y, fmaf(f.z, p2z, noo)));
            const float e = fmaf(b, b, w);
            mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(e - f.w), 31);   // append sign bit, 1 = rejected
        }
        return ~mask << (32 - kCount);
    };
    typedef const __attribute__((address_space(4))) uint32_t* ConstWord;
    const ConstWord force_k = (ConstWord)(sw.force);
    // Chunks whose spheres all have (nearly) the same R are stored normalised: pairs (cx / rho, cz / rho) and one 1 / rho with
    // rho^2 = the largest R of the chunk, and the test becomes q^2 <= 1, q = (cx/rho) ux + (cz/rho) uz - (o.u)/rho.  The
    // threshold is then the same number for every sphere and can sit in a vector register, so a pair costs two FMAs with a
    // scalar operand instead of three (3.7 cycles each here, against 1.65 for an all-VGPR FMA).  Which chunks are stored
    // this way is a bit per chunk, sixteen chunks per word, behind the forced-sphere words.
    typedef float f16v __attribute__((ext_vector_type(16)));
    typedef const __attribute__((address_space(4))) f16v* ConstRec16;
    auto filter_norm = [&](int base, float threshold) -> uint32_t {
        // the chunk's slot holds 32 pairs (cx / rho, cz / rho) in its first 64 floats and 1 / rho in float 64
        const ConstRec16 pairs = (ConstRec16)(prec_k + base);
        const float nc = nou * prec_k[base + 16].x;
        float acc0 = 0.0f, acc1 = 0.0f;
        // Scenes too large for the LDS mirror are also too large for the scalar cache (20 000 spheres: 160 KB of pairs through
        // 16 KB), and there a scalar load is a trip to L2.  Scalar loads come back out of order, so a wave can only wait for all
        // that are outstanding: the next eight spheres are requested right AFTER the wait for the current eight (after their
        // first use) and have the other 31 instructions of the group to arrive; two sets of sixteen scalar registers alternate
        // (20 000-sphere scene 378 -> 357 ms at 32 spp).  Where the records stay in the scalar cache the sixteen extra registers
        // cost more than the overlap gains (487 spheres: 89 -> 95 ms, ten more spilled registers), so that form loads and waits.
        constexpr bool kPipelinedRecords = !kLds;
        auto eight = [&](const f16v& f, int g, auto&& after_first) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float q = fmaf(f[2 * i], ux, fmaf(f[2 * i + 1], uz, nc));
                float verdict;
                asm("v_fma_f32 %0, -%1, %1, %2 clamp" : "=v"(verdict) : "v"(q), "v"(threshold));
                if (g < 2) acc0 = fmaf(acc0, 2.0f, verdict); else acc1 = fmaf(acc1, 2.0f, verdict);
                if (i == 0) after_first();
            }
        };
        if constexpr (kPipelinedRecords) {
            f16v next = pairs[0];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f16v f = next;
                eight(f, g, [&] {
                    __builtin_amdgcn_sched_barrier(0);
                    if (g < 3) next = pairs[g + 1];
                    __builtin_amdgcn_sched_barrier(0);
                });
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f16v f = pairs[g];
                eight(f, g, [] {});
            }
        }
        return ((uint32_t)acc0 << 16) | (uint32_t)acc1;
    };
    float threshold88 = 0x1p88f;       // handed to the FMA as a vector-register operand
    // Matrix-pipe form.  q = (cx ux + cz uz - o.u) / rho of 32 spheres x 32 rays per v_mfma_f32_32x32x8_bf16_1k: every factor
    // is split into a bf16 head and tail (x = xh + xl up to 2^-16 |x|) and the eight products kept are
    //   axh uxh + axl uxh + axh uxl  +  azh uzh + azl uzh + azh uzl  +  ir nh + ir nl        (a = c / rho, ir = 1 / rho a bf16 number),
    // exact in the f32 accumulator, so q is off by < 2^-14 (|ax ux| + |az uz| + |ir o.u|) + the accumulator's roundings; the host
    // widens rho and mfma_g shrinks the normal for 2^-13 of those magnitudes.  Operand layout: lane l supplies row / column l % 32
    // and k = 4 (l / 32) .. + 3.  A comes ready-made from the host (SweepScene::arec); B is built here from the lane's own ray
    // -- k 0-3 in L, k 4-7 in H -- and two half-wave swaps turn (L, H) into the operands for rays 0-31 and for rays 32-63.
    // Result layout: lane l holds column l % 32 and rows 8 (v / 4) + 4 (l / 32) + v % 4, v = 0..15; the host orders a chunk's rows so
    // that these are spheres 16 (l / 32) + v.  Verdict and accumulate are the two all-VGPR FMAs of filter_norm; the chain starts at
    // 128, so that it ends as 2^23 + the sixteen verdict bits, i.e. with those bits in the low half of the float's pattern and no
    // conversion; one more swap hands every lane the two halves of ITS ray's word.
    bf16x4 mb1 = { 0, 0, 0, 0 }, mb2 = { 0, 0, 0, 0 };
    if constexpr (kMfma) {
        if (!active) ux = __builtin_nanf("");
        const uint32_t xh = bf16_bits(ux), zh = bf16_bits(uz), nh = bf16_bits(nou);
        const uint32_t xl = bf16_bits(ux - __uint_as_float(xh)), zl = bf16_bits(uz - __uint_as_float(zh)), nl = bf16_bits(nou - __uint_as_float(nh));
        uint32_t L0 = (xh >> 16) | xh, L1 = (xl >> 16) | zh, H0 = (zh >> 16) | zl, H1 = (nh >> 16) | nl;
        swap_halves(L0, H0); swap_halves(L1, H1);
        mb1 = __builtin_bit_cast(bf16x4, make_uint2(L0, L1));
        mb2 = __builtin_bit_cast(bf16x4, make_uint2(H0, H1));
        asm volatile("" : "+v"(threshold88));
    }
    auto filter_mfma = [&](uint2 araw) -> uint32_t {
        const bf16x4 a = __builtin_bit_cast(bf16x4, araw);
        f16acc zero;
#pragma unroll
        for (int v = 0; v < 16; ++v) zero[v] = 0.0f;
        const f16acc d1 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, mb1, zero, 0, 0, 0);
#ifndef RT_MFMA_SERIAL
        const f16acc d2 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, mb2, zero, 0, 0, 0);
#endif
        float acc1 = 128.0f, acc2 = 128.0f;
        // (the [0, 1] median is folded into the FMA's clamp modifier; written this way, not as inline assembly, the compiler
        // sees the read of a matrix result and keeps the wait states it needs)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc1 = fmaf(acc1, 2.0f, __builtin_amdgcn_fmed3f(fmaf(-d1[v], d1[v], threshold88), 0.0f, 1.0f));
#ifdef RT_MFMA_SERIAL
        __builtin_amdgcn_sched_barrier(0);
        const f16acc d2 = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, mb2, zero, 0, 0, 0);
#endif
#pragma unroll
        for (int v = 0; v < 16; ++v) acc2 = fmaf(acc2, 2.0f, __builtin_amdgcn_fmed3f(fmaf(-d2[v], d2[v], threshold88), 0.0f, 1.0f));
        uint32_t m1 = __float_as_uint(acc1), m2 = __float_as_uint(acc2);
        swap_halves(m1, m2);
        return __builtin_amdgcn_perm(m1, m2, 0x05040100u);       // (low half of m1) << 16 | low half of m2
    };

    T.mark(1);
    // Super-chunks of up to 16 mask words (512 spheres).  The filter only stores each word and notes
    // whether it has a survivor; afterwards one loop per lane walks its surviving bits in ascending
    // sphere order and runs the exact test of sphere.cpp:15-52 with the shrinking upper bound of
    // hittablelist.cpp:21-29 (unfused arithmetic).
    for (int sbase = 0; sbase < sw.n_pad; sbase += 32 * kSuperChunkWords) {
        if constexpr (kTiled) {
            static_assert(32 * kSuperChunkWords == kTileSpheres, "a tile is a super-chunk");
            if (sbase != 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                ++*tiles_done;
            }
            const int next_tile = sbase + kTileSpheres < sw.n_pad ? (sbase >> 9) + 1 : 0;
            stage_operand_tile(sw, lds.arec, next_tile, (*tiles_done + 1u) & 1u);
        }
        uint32_t nz = 0u;          // bit (31 - w) set: word w of this super-chunk has a survivor
        int words = 0;
        int base = sbase;
        const int send = (sw.n_pad - sbase < 32 * kSuperChunkWords) ? sw.n_pad : sbase + 32 * kSuperChunkWords;
        // one flag word covers 16 chunks (512 spheres); this super-chunk's share of it, bit (kSuperChunkWords - 1 - w): word w is stored normalised
        constexpr uint32_t kAllWords = (1u << kSuperChunkWords) - 1u;
        const int flag_shift = kSuperChunkWords * (16 / kSuperChunkWords - 1 - ((sbase / (32 * kSuperChunkWords)) % (16 / kSuperChunkWords)));
        const uint32_t normalised = kPlane && !kMfma ? (force_k[((sw.n_pad + 31) >> 5) + (sbase >> 9)] >> flag_shift) & kAllWords : 0u;
        const int full_words = (send - sbase) >> 5;
        // two loops with one body each (a single loop choosing between the two unrolled bodies per word makes the
        // register allocator spill twenty registers)
        if constexpr (kMfma) {
            // one matrix-pipe chunk per mask word, the operand of the next chunk requested before this one is worked on
            const int words_here = (send - sbase + 31) >> 5;
            const uint2* ap = kTiled ? lds.arec + (size_t)(*tiles_done & 1u) * kTileEntries + (threadIdx.x & 63u)
                                     : (kLds ? lds.arec : sw.arec) + (size_t)(sbase >> 5) * 64 + (threadIdx.x & 63u);
            uint2 a_next = ap[0];
            uint32_t seen = 0u;
            set_wave_priority(0);
#ifdef RT_MFMA_UNROLL
#pragma unroll RT_MFMA_UNROLL
#endif
            for (int w = 0; w < words_here; ++w) {
                const uint2 a_now = a_next;
                if (w + 1 < words_here) a_next = ap[(w + 1) * 64];
                const uint32_t cand = filter_mfma(a_now);
                mask_words[w * 64] = cand;
                seen = seen + seen + (cand != 0u ? 1u : 0u);
            }
            nz = seen << (32 - words_here);
            words = words_here; base = send;
            set_wave_priority(prio_after);
        } else {
            if (normalised != 0u) {
                for (int w = 0; w < full_words; ++w) {
                    if (((normalised >> (kSuperChunkWords - 1 - w)) & 1u) == 0u) continue;
                    const uint32_t cand = filter_norm(sbase + 32 * w, threshold88);
                    mask_words[w * 64] = cand;
                    nz |= cand != 0u ? 0x80000000u >> w : 0u;
                }
            }
            if (normalised != (kAllWords & ~(kAllWords >> full_words))) {
                for (int w = 0; w < full_words; ++w) {
                    if (((normalised >> (kSuperChunkWords - 1 - w)) & 1u) != 0u) continue;
                    const uint32_t cand = filter_bits(sbase + 32 * w, std::integral_constant<int, 32>());
                    mask_words[w * 64] = cand;
                    nz |= cand != 0u ? 0x80000000u >> w : 0u;
                }
            }
            words = full_words; base = sbase + 32 * full_words;
            if (base < send) {         // tail: 8, 16 or 24 records
                const int got = send - base;
                const uint32_t cand = got == 8 ? filter_bits(base, std::integral_constant<int, 8>())
                                    : got == 16 ? filter_bits(base, std::integral_constant<int, 16>()) : filter_bits(base, std::integral_constant<int, 24>());
                base = send;
                mask_words[words * 64] = cand;
                nz |= cand != 0u ? 0x80000000u >> words : 0u;
                ++words;
            }
        }
        // spheres the plane filter does not judge (none in ordinary scenes): their bits are added here, outside the hot loop
        if (kPlane && sw.n_forced > 0) {
            for (int w = 0; w < words; ++w) {
                const uint32_t f = force_k[(sbase >> 5) + w];
                if (f != 0u && active) { mask_words[w * 64] |= f; nz |= 0x80000000u >> w; }
            }
        }
        T.mark(1);
        // Pass A: every survivor of the plane filter goes through the discriminant filter (the 8-FMA form
        // above, on the same per-ray constants and the per-sphere constant k; it never rejects a sphere
        // whose reference discriminant is positive), and the bits of the rejected ones are cleared in
        // place.  Pass B: the reference's own discriminant, roots, range test and running minimum
        // (sphere.cpp:17-50, hittablelist.cpp:21-29) for the two or so per ray that are left, in
        // ascending index order.  Pass A is the long loop (a dozen trips per ray, the wave waits for its
        // slowest lane), so it holds the cheap conservative test; the unfused arithmetic, the square
        // root and the two divisions are paid only in pass B.
        // Every word flagged in nz is non-zero, and padding records never pass the filter, so each bit
        // visited names a real sphere (k < n).
        uint32_t nzb = 0u;
        if (nz != 0u) {
            int w = __builtin_clz(nz);
            nz &= ~(0x80000000u >> w);
            uint32_t cand = mask_words[w * 64], keep = 0u;
            int cbase = sbase + (w << 5);
#if RT_PASSA_ADDR
            // (LDS address of the word's first record, kept as a number the optimiser cannot fold back into the index: a survivor's
            // address is then one shift-and-add)
            typedef const __attribute__((address_space(3))) v4f* LdsRecord;
            const uint32_t lds_geom = kLds ? (uint32_t)(uintptr_t)(LdsRecord)lds.geom : 0u;
            uint32_t rec0 = lds_geom + ((uint32_t)cbase << 4);
            asm volatile("" : "+v"(rec0));
#endif
            for (;;) {
                const int jj = __builtin_clz(cand);
                const uint32_t bit = 0x80000000u >> jj;
                cand ^= bit;                   // the bit is set: a two-cycle xor instead of a four-cycle and-not
                const int k = cbase + jj;
                // without the LDS mirror the discriminant-filter record itself is gathered: it holds the same centre as the exact
                // record and the constant beside it (one 16-byte load per survivor instead of two)
#if RT_PASSA_ADDR
                float4 g;
                if constexpr (kLds) { const v4f q = *(LdsRecord)(uintptr_t)(((uint32_t)jj << 4) + rec0); g = make_float4(q.x, q.y, q.z, q.w); }
                else g = sw.frec[k];
#else
                const float4 g = kLds ? lds.geom[k] : sw.frec[k];
#endif
                const float dk = !kLds ? g.w : kPacked ? g.w : lds.disck[k];
                const float b = fmaf(g.x, hx, fmaf(g.y, hy, fmaf(g.z, hz, nod)));
                // 2 o.c - |o|^2' as (o.c) doubled: four roundings like the filter's chain over 2 o, without keeping 2 o in registers
                // (the compiler recomputed 2 o per survivor rather than hold three more registers across this loop)
                const float ew = fmaf(fmaf(g.x, ray.o.x, fmaf(g.y, ray.o.y, g.z * ray.o.z)), 2.0f, noo);
                const float e = fmaf(b, b, ew);
                if (!(e - dk < 0.0f)) keep |= bit;
                if (cand == 0u) {
                    mask_words[w * 64] = keep;
                    nzb |= keep ? 0x80000000u >> w : 0u;
                    if (nz == 0u) break;
                    w = __builtin_clz(nz);
                    nz &= ~(0x80000000u >> w);
                    cand = mask_words[w * 64];
                    keep = 0u;
                    cbase = sbase + (w << 5);
#if RT_PASSA_ADDR
                    rec0 = lds_geom + ((uint32_t)cbase << 4);
                    asm volatile("" : "+v"(rec0));
#endif
                }
            }
        }
        T.mark(2);
        if (nzb != 0u) {
            int w = __builtin_clz(nzb);
            nzb &= ~(0x80000000u >> w);
            uint32_t cand = mask_words[w * 64];
            int cbase = sbase + (w << 5);
            for (;;) {
                const int jj = __builtin_clz(cand);
                cand ^= 0x80000000u >> jj;
                const int k = cbase + jj;
                const float4 g = kLds ? lds.geom[k] : sw.geom[k];
                const V3 oc = ray.o - mk(g.x, g.y, g.z);
                const float b = dot(oc, ray.d);
                const float cc = dot(oc, oc) - exact_r2<kLds && kPacked>(g, lds, k);
                const float desc = b * b - qa * cc;
                const float sq = sqrtf(desc);
                float temp = (-b - sq) / qa;
                if (!(temp < closest && temp > tmin)) temp = (-b + sq) / qa;
                if (desc > 0.0f && temp < closest && temp > tmin) { closest = temp; hit = k; }
                if (cand == 0u) {
                    if (nzb == 0u) break;
                    w = __builtin_clz(nzb);
                    nzb &= ~(0x80000000u >> w);
                    cand = mask_words[w * 64];
                    cbase = sbase + (w << 5);
                }
            }
        }
        T.mark(3);
    }
}

// The same closest-hit query for a wave that holds only a few live paths (the tail of a launch, when the work
// counter has run dry): instead of one ray per lane and a loop over the spheres, the live rays are taken one
// at a time and the SPHERES are spread over the 64 lanes (sphere k on lane k mod 64), followed by a wave
// reduction.  A sweep for a sparse wave costs as many issue slots as one for a full wave, and a lone wave issues
// slowly, so this shortens the bounces of the last long paths several times over.  The result is the same
// (sphere, t): HittableList::Hit's answer is the smallest accepted root over all spheres with ties going to the
// lower index (hittablelist.cpp:21-29 with sphere.cpp:27-50: for one sphere the accepted root under a running
// bound c is its first root above tmin, if below c), which does not depend on how the spheres are partitioned.
#ifndef RT_FEW_PATHS
#define RT_FEW_PATHS 8
#endif
// depth of the deepest path a wave holds at which it is raised to issue priority 1, 2, 3 (4/8/16, 2/4/8, 12/20/32 and
// 16/24/36 measure within the run-to-run spread of this choice, tools/ab_prio.sh)
#ifndef RT_PRIO_1
#define RT_PRIO_1 8
#define RT_PRIO_2 16
#define RT_PRIO_3 32
#endif
// The end of a launch shared by a block (k_trace, matrix-pipe variants): its waves pool their last paths once they hold no more than
// this many together (0: every wave drains its own)
#ifndef RT_TAIL_PATHS
#define RT_TAIL_PATHS 48
#endif
constexpr int kTailPaths = RT_TAIL_PATHS;
constexpr int kTailCtlBytes = 128;          // control words of the shared end, at the start of the block's LDS
constexpr int kTailFields = 10;             // words per pooled path: origin, direction, pixel key, sample, depth | stack owner << 8, sample-buffer slot
constexpr int kFewPaths = RT_FEW_PATHS;     // 2, 4 and 8 measure alike on the 1200x800 final scene at 10 spp, 16 and 24 are 4-6 % slower (tools/ab_few.sh)
template <bool kLds>
__device__ __forceinline__ void closest_hit_few(const RayD& ray, float tmin, const SweepScene& sw, const ExactLds& lds,
                                                unsigned long long live, int lane, float& closest, int& hit)
{
    // A lone wave issues one dependent instruction every few cycles, so what a bounce costs here is the LENGTH of its dependency
    // chains, not the number of instructions: the discriminants of four spheres per lane are formed side by side (four independent
    // chains, 256 spheres per batch), the square root and the two divisions are paid only for lanes whose line meets a sphere, and
    // the wave minimum is taken over the few lanes that found a root by reading them out one by one (scalar), not by a
    // six-step butterfly of cross-lane moves.
    constexpr int kPerLane = 4;
    while (live != 0ull) {
        const int src = __builtin_ctzll(live);
        live &= live - 1ull;
        RayD r;
        r.o = mk(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ray.o.x), src)),
                 __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ray.o.y), src)),
                 __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ray.o.z), src)));
        r.d = mk(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ray.d.x), src)),
                 __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ray.d.y), src)),
                 __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ray.d.z), src)));
        const float qa = dot(r.d, r.d);
        float best = FLT_MAX; int best_k = -1;
        for (int base = 0; base < sw.n; base += 64 * kPerLane) {
            float b[kPerLane], desc[kPerLane];
#pragma unroll
            for (int j = 0; j < kPerLane; ++j) {
                const int k = base + 64 * j + lane;
                b[j] = 0.0f; desc[j] = -1.0f;
                if (k < sw.n) {
                    const float4 g = kLds ? lds.geom[k] : sw.geom[k];
                    const V3 oc = r.o - mk(g.x, g.y, g.z);
                    b[j] = dot(oc, r.d);
                    const float cc = dot(oc, oc) - exact_r2<kLds>(g, lds, k);
                    desc[j] = b[j] * b[j] - qa * cc;
                }
            }
            // ascending sphere index inside the lane, strict comparisons: equal roots stay with the lower index.  The roots of a
            // group of 64 spheres are worked out only when the line meets one of them (three or four of a scene's eight groups per
            // ray, the ground sphere's among them): the square root and the two divisions are sixty instructions that the whole
            // wave issues whether one lane needs them or all
#pragma unroll
            for (int j = 0; j < kPerLane; ++j) {
                if (__any(desc[j] > 0.0f)) {
                    if (desc[j] > 0.0f) {
                        const float sq = sqrtf(desc[j]);
                        float temp = (-b[j] - sq) / qa;
                        if (!(temp < best && temp > tmin)) temp = (-b[j] + sq) / qa;
                        if (temp < best && temp > tmin) { best = temp; best_k = base + 64 * j + lane; }
                    }
                }
            }
        }
        // wave minimum of (t, index) over the lanes that hold a root: smaller t wins, equal t goes to the lower index
        unsigned long long cand = __ballot(best_k >= 0);
        float wt = FLT_MAX; int wk = -1;
        while (cand != 0ull) {
            const int l = __builtin_ctzll(cand);
            cand &= cand - 1ull;
            const float t = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, best), l));
            const int k = __builtin_amdgcn_readlane(best_k, l);
            if (t < wt || (t == wt && (unsigned)k < (unsigned)wk)) { wt = t; wk = k; }
        }
        if (lane == src) { closest = wt; hit = wk; }
    }
}

// ============================================================================ closest hit through the sphere hierarchy
// RT_MODE_BVH: the same closest hit as the brute-force sweep, found by walking the hierarchy of rt_bvh.hpp per lane.
// Every sphere that is tested goes through the reference's own arithmetic (sphere.cpp:15-52); the hierarchy only
// decides which spheres can be skipped, so the box test has to be conservative against that arithmetic:
//   * the reference reports a hit when its float discriminant b^2 - a c is positive.  With u = 2^-24, counting the
//     roundings of sphere.cpp:17-25 (oc: u per component; b: 4 u |oc||d|, so b^2: 9 u a |oc|^2; oc.oc - r^2:
//     5 u |oc|^2 + 2 u r^2; a: 3 u a; the product a c: a (9 u |oc|^2 + 6 u r^2); the difference: u a |oc|^2), the
//     computed value is off by less than a (19 u |o - c|^2 + 6 u r^2), so the line may pass up to
//     sqrt(r^2 + 19 u |o - c|^2 + 6 u r^2) from the centre.  A node's centre box is therefore inflated by
//     sqrt(rmax^2 + 24 u (D^2 + rmax^2)), D = distance from the ray origin to the farthest corner of the box
//     (>= |o - c| for every sphere below);
//   * the slab test's own rounding is covered by a further relative 1e-5 and an absolute 1e-6 (1 + |o| + |box|),
//     the box's part of which the builder adds to the radius stored in the node.
// The closest hit over a set of spheres is the smallest accepted root, equal roots going to the lower index
// (hittablelist.cpp:21-29 keeps the first of equal ones and visits in index order); spheres are met here in
// tree order, so the tie is resolved explicitly.
constexpr float kBvhSlack = 24.0f * 5.9604645e-08f;

template <bool kLds>
__device__ __forceinline__ void bvh_test_sphere(const RayD& ray, float qa, float tmin, const BvhScene& bv, const ExactLds& lds, int k, float& closest, int& hit)
{
    const float4 g = kLds ? lds.geom[k] : bv.geom[k];       // plain form: (centre, r*r)
    const V3 oc = ray.o - mk(g.x, g.y, g.z);
    const float b = dot(oc, ray.d);
    const float cc = dot(oc, oc) - g.w;
    const float desc = b * b - qa * cc;
    if (desc > 0.0f) {
        const float sq = sqrtf(desc);
        float temp = (-b - sq) / qa;
        if (!(temp > tmin)) temp = (-b + sq) / qa;          // the first root above tmin is the one a running bound can accept
        if (temp > tmin && (temp < closest || (temp == closest && k < hit))) { closest = temp; hit = k; }
    }
}

// Conservative ray / inflated-centre-box test of one node (see above).  Returns the entry parameter in t_in.
__device__ __forceinline__ bool bvh_enter(const RayD& ray, float ix, float iy, float iz, float ray_margin, float t_low, float closest,
                                          const float4 A, const float4 B, float& t_in)
{
    const float fx = fmaxf(fabsf(ray.o.x - A.x), fabsf(ray.o.x - B.x));
    const float fy = fmaxf(fabsf(ray.o.y - A.y), fabsf(ray.o.y - B.y));
    const float fz = fmaxf(fabsf(ray.o.z - A.z), fabsf(ray.o.z - B.z));
    const float far2 = fmaf(fx, fx, fmaf(fy, fy, fz * fz));
    const float r2 = A.w * A.w;
    // A.w already carries the node's share of the absolute margin (1e-6 (|cmin| + |cmax|), added by the builder);
    // the hardware square root (1 ulp) is inside the relative 1e-5
    const float pad = fmaf(__builtin_amdgcn_sqrtf(fmaf(kBvhSlack, far2 + r2, r2)), 1.00001f, ray_margin);
    const float ax = (A.x - pad - ray.o.x) * ix, bx = (B.x + pad - ray.o.x) * ix;
    const float ay = (A.y - pad - ray.o.y) * iy, by = (B.y + pad - ray.o.y) * iy;
    const float az = (A.z - pad - ray.o.z) * iz, bz = (B.z + pad - ray.o.z) * iz;
    const float t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    const float t1 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    t_in = t0;
    return t0 <= t1 && t1 >= t_low && t0 <= closest;          // false for a NaN ray: it hits nothing in the reference either
}

// pending: this lane's column of kBvhStack LDS words (entry e at pending[e * 64]).  The walk handles the two
// children of a node together: leaves among them are tested at once, of two inner children the nearer is entered
// and the other one's child pair is left pending (it is re-examined against the then-current closest hit when popped).
template <bool kLds>
__device__ __forceinline__ void closest_hit_bvh(const RayD& ray, float tmin, const BvhScene& bv, const ExactLds& lds, uint32_t* pending, float& closest, int& hit)
{
    const float qa = dot(ray.d, ray.d);
    for (int j = 0; j < bv.n_big; ++j) bvh_test_sphere<kLds>(ray, qa, tmin, bv, lds, (int)bv.big[j], closest, hit);
    const float ix = 1.0f / ray.d.x, iy = 1.0f / ray.d.y, iz = 1.0f / ray.d.z;
    const float omag = 1e-6f * (1.0f + fabsf(ray.o.x) + fabsf(ray.o.y) + fabsf(ray.o.z));     // the ray's share of the absolute margin
    const float t_low = tmin < 0.0f ? -FLT_MAX : 0.0f;
    auto test_leaf = [&](int link) {
        const int code = -(link + 1), first = code >> 3, count = (code & 7) + 1;
        for (int q = 0; q < count; ++q) bvh_test_sphere<kLds>(ray, qa, tmin, bv, lds, (int)bv.leaf_index[first + q], closest, hit);
    };
    const int root_link = __float_as_int(bv.nodes[1].w);
    if (root_link < 0) {                    // at most one leaf in the whole tree
        float t_in;
        if (bvh_enter(ray, ix, iy, iz, omag, t_low, closest, bv.nodes[0], bv.nodes[1], t_in)) test_leaf(root_link);
        return;
    }
    int sp = 0; uint32_t pair = (uint32_t)root_link;
    for (;;) {
        const float4 A0 = bv.nodes[2u * pair], B0 = bv.nodes[2u * pair + 1u], A1 = bv.nodes[2u * pair + 2u], B1 = bv.nodes[2u * pair + 3u];
        float t_in0, t_in1;
        bool in0 = bvh_enter(ray, ix, iy, iz, omag, t_low, closest, A0, B0, t_in0);
        bool in1 = bvh_enter(ray, ix, iy, iz, omag, t_low, closest, A1, B1, t_in1);
        const int link0 = __float_as_int(B0.w), link1 = __float_as_int(B1.w);
        if (in0 && link0 < 0) { test_leaf(link0); in0 = false; }
        if (in1 && link1 < 0) { test_leaf(link1); in1 = false; }
        in0 = in0 && t_in0 <= closest;      // the leaves may have brought the closest hit nearer
        in1 = in1 && t_in1 <= closest;
        if (in0 && in1) {
            const bool first0 = t_in0 <= t_in1;
            pending[sp * 64] = (uint32_t)(first0 ? link1 : link0); ++sp;
            pair = (uint32_t)(first0 ? link0 : link1);
            continue;
        }
        if (in0) { pair = (uint32_t)link0; continue; }
        if (in1) { pair = (uint32_t)link1; continue; }
        if (sp == 0) break;
        --sp; pair = pending[sp * 64];
    }
}

// ============================================================================ shading step of Color()
// One level of the recursion main.cpp:13-39 for a ray whose closest hit is known.  Returns true when the
// path has ended (sample = its radiance); otherwise ray/depth are advanced and the hit index is pushed on
// the lane's stack (stack.at(depth)).
// kNoStack: for a camera ray (depth 0) whose caller records the hit index itself; nothing is pushed and there is nothing to replay.
// The hit-index stack of a path, two forms: a per-lane pointer with a stride between levels, and -- for the persistent kernel,
// where a 64-bit pointer per lane is two registers held across the whole main loop -- a base every lane shares plus the lane's
// 32-bit column (level k of column c at base[k * stride + c]; the address is then a scalar base and one 32-bit vector offset).
struct StridedStack {
    uint16_t* p; size_t stride;
    __device__ __forceinline__ uint16_t& at(int k) const { return p[(size_t)k * stride]; }
};
struct ColumnStack {
    uint16_t* base; uint32_t column, stride;        // stride * RT_MAX_DEPTH + column < 2^31: the byte offset is formed in 32 bits
    __device__ __forceinline__ uint16_t& at(int k) const { return *reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(base) + (((uint32_t)k * stride + column) << 1)); }
};
template <bool kLds, bool kNoStack = false, class Uniforms = KeyedUniforms, class Stack = StridedStack>
__device__ __forceinline__ bool shade_step(RayD& ray, int& depth, float closest, int hit, uint32_t pix, uint32_t smp, int max_depth,
                                           Uniforms& U, const float4* s_geom, const float4* s_mat, const Aux* s_aux,
                                           const float4* g_geom, const float4* g_mat, const Aux* g_aux,
                                           const Stack& stack, V3& sample, SecTimer& T)
{
    sample = mk(0.0f, 0.0f, 0.0f);
    bool done = true;
    if (hit >= 0) {
        const float4 g = kLds ? s_geom[hit] : g_geom[hit];
        const float4 m = kLds ? s_mat[hit] : g_mat[hit];
        const Aux x = kLds ? s_aux[hit] : g_aux[hit];
        const V3 p = ray.o + closest * ray.d;
        const V3 nrm = (p - mk(g.x, g.y, g.z)) / x.radius;
        T.mark(5);
        if (depth < max_depth) {
            V3 att; RayD out;
            U.slot(pix, smp, (uint32_t)depth + 1u);
            if (scatter(x.type, mk(m.x, m.y, m.z), m.w, ray, p, nrm, U, att, out)) {
                if (!kNoStack) stack.at(depth) = (uint16_t)hit;
                ++depth; ray = out; done = false;
            }
        }
        T.mark(6);
    } else {
        sample = sky(ray.d);
        // The attenuation chain, innermost first.  Two levels per round: their stack entries are requested together, then their
        // albedos, then the products are formed in order -- one memory latency per round instead of one per level (a level below
        // the first re-reads level 0 and is not multiplied in).  Config 3 with 1 / 2 / 4 / 8 levels per round: 68.5 / 68.3 / 68.5 / 69.0 ms.
#ifndef RT_REPLAY_GROUP
#define RT_REPLAY_GROUP 2
#endif
#pragma clang loop unroll(disable)
        for (int k = kNoStack ? -1 : depth - 1; k >= 0; k -= RT_REPLAY_GROUP) {
            int h[RT_REPLAY_GROUP];
#pragma unroll
            for (int j = 0; j < RT_REPLAY_GROUP; ++j) h[j] = stack.at(k - j >= 0 ? k - j : 0);
            float4 m[RT_REPLAY_GROUP];
#pragma unroll
            for (int j = 0; j < RT_REPLAY_GROUP; ++j) m[j] = kLds ? s_mat[h[j]] : g_mat[h[j]];
#pragma unroll
            for (int j = 0; j < RT_REPLAY_GROUP; ++j) if (k - j >= 0) sample = mk(m[j].x, m[j].y, m[j].z) * sample;
        }
    }
    return done;
}

// Work item -> (pixel key, sample, pixel coordinates).  Items are numbered tile by tile, then sample by
// sample, then pixel by pixel inside the 8x8 tile: item = (tile * s_count + s_rel) * 64 + row_in_tile * 8 +
// col_in_tile, so 512 consecutive items are eight samples of ONE tile.  Returns false for the padding pixels
// of edge tiles.
//
// tile_order (optional) is a permutation of the tiles: work position `tile` is image tile tile_order[tile], so
// that tiles can be WORKED on in another order than the image's.  The sample buffer is kept in work order -- `store`, the slot
// of the item's sample, is the item number itself -- and the resolve kernels look a tile's work position up.
//
// A tile share (rt_render_tiles_device): the call works on every share_step-th position of the work order, starting at
// share_first -- work tile w is position share_first + w * share_step.  That is
// how one image is dealt out to several GPUs tile by tile: every share gets the same mix of expensive and cheap tiles.
struct ItemGeometry {
    int width, height, band_rows, band_first, band_step, owned_rows, tiles_x; uint32_t Q; int s_begin, s_count; const uint32_t* tile_order;
    int share_first = 0, share_step = 1;
};
__device__ __forceinline__ uint32_t image_tile(const ItemGeometry& ig, uint32_t work_tile)
{
    const uint32_t pos = (uint32_t)ig.share_first + work_tile * (uint32_t)ig.share_step;
    return ig.tile_order ? ig.tile_order[pos] : pos;
}
// The same numbering for a whole row of 64 items at once: `row` = item >> 6 (one tile at one sample) is the same for every lane of
// a wave that holds such a row, `l` = item & 63 is the lane's pixel of the tile.  With `row` in a scalar register the two
// divisions by the sample count and the tile-row length, the look-up of the tile order and the band arithmetic of the tile's first
// pixel row run once per wave on the scalar unit instead of once per lane with quarter-rate integer multiplies.
__device__ __forceinline__ bool decode_item_row(const ItemGeometry& ig, uint32_t row, uint32_t l, uint32_t& pix, uint32_t& smp, int& i, int& j)
{
    const uint32_t work_tile = row / (uint32_t)ig.s_count, s_rel = row - work_tile * (uint32_t)ig.s_count;
    const uint32_t tile = image_tile(ig, work_tile);
    const int ty = (int)(tile / (uint32_t)ig.tiles_x), tx = (int)(tile - (uint32_t)ty * (uint32_t)ig.tiles_x);
    i = tx * 8 + (int)(l & 7u);
    const int r = ty * 8 + (int)(l >> 3);
    // owned_row_to_y2 for r = ty * 8 + 0..7: the band of the tile's first row once, then at most one band boundary inside the
    // tile when a band has eight rows or more (a division per lane otherwise)
    int band = (ty * 8) / ig.band_rows, in_band = ty * 8 - band * ig.band_rows + (int)(l >> 3);
    if (ig.band_rows >= 8) { if (in_band >= ig.band_rows) { in_band -= ig.band_rows; ++band; } }
    else { const int over = in_band / ig.band_rows; band += over; in_band -= over * ig.band_rows; }
    const int y = (ig.band_first + band * ig.band_step) * ig.band_rows + in_band;
    if (!(i < ig.width && r < ig.owned_rows && y < ig.height)) return false;
    j = ig.height - 1 - y;
    pix = (uint32_t)j * (uint32_t)ig.width + (uint32_t)i;
    smp = (uint32_t)ig.s_begin + s_rel;
    return true;
}
__device__ __forceinline__ bool decode_item(const ItemGeometry& ig, uint32_t item, uint32_t& pix, uint32_t& smp, int& i, int& j, uint32_t& store)
{
    const uint32_t ts = item >> 6, l = item & 63u;
    const uint32_t work_tile = ts / (uint32_t)ig.s_count, s_rel = ts - work_tile * (uint32_t)ig.s_count;
    const uint32_t tile = image_tile(ig, work_tile);
    store = item;
    const int ty = (int)(tile / (uint32_t)ig.tiles_x), tx = (int)(tile - (uint32_t)ty * (uint32_t)ig.tiles_x);
    i = tx * 8 + (int)(l & 7u);
    const int r = ty * 8 + (int)(l >> 3);
    const int y = owned_row_to_y2(ig.band_rows, ig.band_first, ig.band_step, r < ig.owned_rows ? r : 0);
    if (!(i < ig.width && r < ig.owned_rows && y < ig.height)) return false;
    j = ig.height - 1 - y;
    pix = (uint32_t)j * (uint32_t)ig.width + (uint32_t)i;
    smp = (uint32_t)ig.s_begin + s_rel;
    return true;
}
