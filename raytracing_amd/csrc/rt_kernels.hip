// HIP kernels (gfx950) and the extern "C" layer of librt_amd.so.  See include/rt_amd.h.
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>
#include <vector>
#include "rt_internal.hpp"
#include "rt_bvh.hpp"

// ============================================================================ device data layout
// Scene records in HBM (built once per upload) and, for the megakernel, mirrored into LDS:
//   geom[i] = (cx, cy, cz, r*r)       16 B  -- everything the sweep touches
//   mat[i]  = (ar, ag, ab, param)     16 B  -- read once per hit
//   aux[i]  = (radius, type)           8 B  -- read once per hit (normal = (p - c) / radius)

struct RenderArgs {
    const float4* geom; const float4* mat; const Aux* aux; int n;
    CameraD cam;
    int width, height, spp, max_depth; float tmin;
    uint32_t k0, k1;
    int band_rows, band_first, band_step, owned_rows;
    float* rgb_mean; int32_t* irgb;
    unsigned long long* ray_counter;
};

static constexpr int kBlock = 256;          // 4 waves; each wave owns an 8x8 pixel tile
static constexpr int kLdsBytesPerSphere = 40;
static constexpr int kMaxLdsSpheres = 1536; // 60 KB of LDS per block -> at least two blocks per CU

// Compact owned row r -> image row y (from the top).
__device__ __forceinline__ int owned_row_to_y2(int band_rows, int band_first, int band_step, int r)
{
    const int k = r / band_rows;
    return (band_first + k * band_step) * band_rows + (r - k * band_rows);
}
__device__ __forceinline__ int owned_row_to_y(const RenderArgs& a, int r)
{
    const int k = r / a.band_rows;
    return (a.band_first + k * a.band_step) * a.band_rows + (r - k * a.band_rows);
}

// Camera ray of (pixel, sample): jitter from words 0,1 of group 0 (main.cpp:126-127), lens sample by
// rejection from the words lens_words() names, then Camera::GetRay (camera.cpp:46-53).
__device__ __forceinline__ RayD camera_ray(const CameraD& cam, uint32_t pix, uint32_t smp, int i, int j, float fnx, float fny, uint32_t k0, uint32_t k1)
{
    U4 blk = philox4x32_10(pix, smp, 0u, 0u, k0, k1);
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
            if (t & 1u) blk = philox4x32_10(pix, smp, 0u, (t + 1u) >> 1, k0, k1);
        }
    }
    return get_ray(cam, fu, fv, dx, dy);
}

// ============================================================================ megakernel
// One lane per pixel.  A lane walks its pixel's samples in order and, within a sample, the
// bounces of Color() (main.cpp:13-39) iteratively; when a path ends the lane starts its next
// sample at once, so every trip of the outer loop issues one closest-hit sweep for every lane
// that still has work.  The attenuation product of the recursion, a0*(a1*(...*sky)), is formed
// innermost-first from a per-lane stack of hit indices so that it rounds exactly as the
// reference's recursion does.
template <bool kLds>
__global__ void __launch_bounds__(kBlock) k_render_mega(const RenderArgs a)
{
    extern __shared__ float4 smem[];
    float4* s_geom = smem;
    float4* s_mat = smem + a.n;
    Aux* s_aux = reinterpret_cast<Aux*>(smem + 2 * a.n);
    __shared__ unsigned long long s_rays, s_misses;
    if (threadIdx.x == 0) { s_rays = 0; s_misses = 0; }
    if (kLds) {
        for (int i = threadIdx.x; i < a.n; i += kBlock) { s_geom[i] = a.geom[i]; s_mat[i] = a.mat[i]; s_aux[i] = a.aux[i]; }
    }
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
    const int r = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
    const int y = owned_row_to_y(a, r < a.owned_rows ? r : 0);
    const bool valid = (i < a.width) && (r < a.owned_rows) && (y < a.height);

    uint32_t rays = 0, misses = 0;
    if (valid) {
        const int j = a.height - 1 - y;
        const uint32_t pix = (uint32_t)j * (uint32_t)a.width + (uint32_t)i;
        KeyedUniforms U; U.k0 = a.k0; U.k1 = a.k1; U.slot(pix, 0, 0);
        uint16_t stack[RT_MAX_DEPTH];
        V3 col = mk(0.0f, 0.0f, 0.0f);
        RayD ray; int depth = 0; int smp = 0;
        const float fnx = (float)a.width, fny = (float)a.height;

        auto next_camera_ray = [&](int s) { ray = camera_ray(a.cam, pix, (uint32_t)s, i, j, fnx, fny, a.k0, a.k1); depth = 0; };
        next_camera_ray(0);

        while (smp < a.spp) {
            // ---- HittableList::Hit (hittablelist.cpp:13-31) over Sphere::Hit (sphere.cpp:15-52)
            float closest = FLT_MAX; int hit = -1;
            const float qa = dot(ray.d, ray.d);
            for (int k = 0; k < a.n; ++k) {
                const float4 g = kLds ? s_geom[k] : a.geom[k];
                const V3 oc = ray.o - mk(g.x, g.y, g.z);
                const float b = dot(oc, ray.d);
                const float cc = dot(oc, oc) - g.w;
                const float desc = b * b - qa * cc;
                if (desc > 0.0f) {
                    const float sq = sqrtf(desc);
                    float temp = (-b - sq) / qa;
                    if (!(temp < closest && temp > a.tmin)) temp = (-b + sq) / qa;
                    if (temp < closest && temp > a.tmin) { closest = temp; hit = k; }
                }
            }
            ++rays;
            V3 sample; bool done;
            if (hit >= 0) {
                const float4 g = kLds ? s_geom[hit] : a.geom[hit];
                const float4 m = kLds ? s_mat[hit] : a.mat[hit];
                const Aux x = kLds ? s_aux[hit] : a.aux[hit];
                const V3 p = ray.o + closest * ray.d;
                const V3 nrm = (p - mk(g.x, g.y, g.z)) / x.radius;
                V3 att; RayD out; bool cont = false;
                if (depth < a.max_depth) {
                    U.slot(pix, (uint32_t)smp, (uint32_t)depth + 1u);
                    cont = scatter(x.type, mk(m.x, m.y, m.z), m.w, ray, p, nrm, U, att, out);
                }
                if (cont) { stack[depth] = (uint16_t)hit; ++depth; ray = out; done = false; }
                else { sample = mk(0.0f, 0.0f, 0.0f); done = true; }
            } else {
                ++misses;
                sample = sky(ray.d);
                for (int k = depth - 1; k >= 0; --k) {
                    const float4 m = kLds ? s_mat[stack[k]] : a.mat[stack[k]];
                    sample = mk(m.x, m.y, m.z) * sample;
                }
                done = true;
            }
            if (done) {
                col = col + sample;
                ++smp;
                if (smp < a.spp) next_camera_ray(smp);
            }
        }
        const float inv = 1.0f / (float)a.spp;          // Vec3::operator/=(float), vec3.cpp:109-118
        const float mr = col.x * inv, mg = col.y * inv, mb = col.z * inv;
        const size_t o = ((size_t)r * a.width + i) * 3;
        if (a.rgb_mean) { a.rgb_mean[o] = mr; a.rgb_mean[o + 1] = mg; a.rgb_mean[o + 2] = mb; }
        if (a.irgb) { a.irgb[o] = quantise(mr); a.irgb[o + 1] = quantise(mg); a.irgb[o + 2] = quantise(mb); }
    }
    if (rays) { atomicAdd(&s_rays, (unsigned long long)rays); atomicAdd(&s_misses, (unsigned long long)misses); }
    __syncthreads();
    if (threadIdx.x == 0 && s_rays) { atomicAdd(a.ray_counter, s_rays); atomicAdd(a.ray_counter + 1, s_misses); }
}

// ============================================================================ persistent megakernel
// Work item = one sample of one pixel.  Items are numbered tile by tile over 8x8 pixel tiles,
//   item = (tile * s_count + s_rel) * 64 + (row_in_tile * 8 + col_in_tile),
// so the 64 items a fresh wave picks up are one tile at one sample (coherent primary rays) and
// both this kernel's stores and the resolve kernel's loads of the sample buffer are coalesced.
// A wave draws items from a chunk it reserves with one atomic on a global counter; whenever a
// lane's path ends it is handed the next item (ballot + prefix count), so every trip of the main
// loop runs one closest-hit sweep with all lanes busy until the work runs out.
//
// The sweep (closest_hit) is split in two:
//   filter  (hot, branch-free, records as scalar operands): a conservative sign test per ray-sphere
//            pair -- by default the distance of the centre to the vertical plane through the ray
//            (3 FMAs), alternatively the whole discriminant expanded about the world origin (8 FMAs) --
//            with a per-sphere and per-ray slack that dominates every rounding error of both this
//            form and the reference's.  The sign bits of 32 spheres are shifted into one register.
//   exact   (cold): the few spheres whose bit says "maybe" are evaluated with the reference's own
//            unfused arithmetic (sphere.cpp:15-52); the accepted root of each and the running minimum
//            follow hittablelist.cpp:13-31, so the chosen sphere and t are bit-identical.
// A wave left with only a few live paths (the end of a launch) switches to closest_hit_few.
// Device view of the sphere hierarchy (rt_bvh.hpp) for RT_MODE_BVH.
struct BvhScene {
    const float4* nodes; const uint32_t* leaf_index; const uint32_t* big; int n_big;
    const float4* geom;
};
struct TraceArgs {
    const float4* prec;      // plane-filter records (cx, cz, R_i = r^2 + slack_i, 0), same padding as frec
    float plane_s0, plane_rmin;   // plane filter: constant part of the per-ray slack, smallest R_i
    const float4* frec;      // discriminant-filter records (cx, cy, cz, |c|^2 - r^2 - slack), padded to a multiple of 8
    const float4* geom; const float4* mat; const Aux* aux; int n, n_pad;
    CameraD cam;
    int width, height, max_depth; float tmin;
    uint32_t k0, k1;
    int band_rows, band_first, band_step, owned_rows;
    int tiles_x; uint32_t Q;              // padded pixel count = tiles * 64
    int s_begin, s_count;                 // sample batch
    uint32_t total_items;                 // Q * s_count  (< 2^31)
    int chunk_items;                      // items a wave reserves per atomic: a multiple of 64, at most kChunkItems
    uint32_t resident_waves;              // waves of this launch that run at the same time
    uint32_t min_chunk_items;             // smallest chunk taken near the end of the work (a multiple of 64)
    float* samples;                       // [total_items][3]
    unsigned long long* ray_counter;
    unsigned int* work_counter;
    float* staging;                       // per wave: kStagingRows rows (o, d, pixel, sample, sample-buffer slot) of kChunkItems pre-generated camera rays
    const uint32_t* tile_order;           // work order of the tiles (see decode_item); may be null
    BvhScene bvh;                         // RT_MODE_BVH only
    uint16_t* stacks; uint32_t stack_stride;   // hit-index stacks in HBM: entry k of lane g at stacks[k * stack_stride + g]
};

// Diagnostic build (-DRT_SECTION_TIMERS): every wave accumulates s_memtime ticks per section of the
// main loop into g_section_ticks.  Never enabled in the shipped library (mark() is empty there).
#ifdef RT_WAVE_TIMELINE
// Diagnostic build (-DRT_WAVE_TIMELINE), per wave of the last k_trace launch: s_memrealtime (100 MHz, common to all XCDs) at entry, when the wave first found the work counter dry, at exit;
// and how many trips of the main loop it made after that point
__device__ unsigned long long g_wave_times[8][8192];      // rows 4, 5: trips made with the sparse sweep, ticks spent in those trips (whole trip); 6: first sweep starts; 7: first trip ends
#endif
#ifdef RT_SECTION_TIMERS
__device__ unsigned long long g_section_ticks[8];
struct SecTimer {
    unsigned long long acc[8]; unsigned long long last;
    __device__ __forceinline__ void start() { for (int k = 0; k < 8; ++k) acc[k] = 0; last = __builtin_amdgcn_s_memtime(); }
    __device__ __forceinline__ void mark(int idx) { const unsigned long long t = __builtin_amdgcn_s_memtime(); acc[idx] += t - last; last = t; }
    __device__ __forceinline__ void flush(int lane) { if (lane == 0) for (int k = 0; k < 8; ++k) atomicAdd(&g_section_ticks[k], acc[k]); }
};
#else
struct SecTimer {
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void flush(int) {}
};
#endif
static constexpr int kChunkItems = 512;
static constexpr int kStagingRows = 9;
static constexpr int kTraceBlock = 512;         // 8 waves share one LDS copy of the scene records
__device__ __forceinline__ float staged_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static constexpr int kSuperChunkWords = 16;                     // 32-sphere mask words kept per lane between exact passes (512 spheres)
static constexpr float kRaySlack = 160.0f * 5.9604645e-08f;   // per-ray slack, relative to |o|^2
static constexpr float kPlaneRaySlack = 128.0f * 5.9604645e-08f;

// ============================================================================ closest-hit sweep
// HittableList::Hit (hittablelist.cpp:13-31) over Sphere::Hit (sphere.cpp:15-52) for one ray per lane,
// brute force over all spheres, as filter + exact (see k_trace).  Shared by the persistent megakernel and
// the wavefront intersect kernel.  mask_words: this lane's column of kSuperChunkWords LDS words.
struct SweepScene {
    const float4* frec;      // disc-filter records (cx, cy, cz, |c|^2 - r^2 - slack), padded to a multiple of 8
    const float4* prec;      // plane-filter records (cx, cz, R_i = r^2 + slack_i, 0), same padding
    float plane_s0;          // constant part of the plane filter's per-ray slack
    float plane_rmin;        // smallest R_i over the non-degenerate spheres
    const float4* geom;      // exact records in HBM (used when they are not mirrored in LDS)
    int n, n_pad;
};

template <bool kLds, bool kPlane, bool kRecLds>
__device__ __forceinline__ void closest_hit(const RayD& ray, float tmin, const SweepScene& sw, const float4* s_geom, const float4* s_prec,
                                            uint32_t* mask_words, float& closest, int& hit, SecTimer& T)
{
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
        const float kappa = sqrtf(k2) * (1.0f - 4.0f * 5.9604645e-08f);
        ux *= kappa; uz *= kappa;
        nou = -fmaf(ray.o.x, ux, ray.o.z * uz);
    }
    // Filter records are read through the constant address space: the index is wave-uniform,
    // so each record arrives by scalar load in SGPRs and feeds the FMAs as a scalar operand.
    typedef float f4v __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) f4v* ConstRec;
    const ConstRec frec_k = (ConstRec)(sw.frec);
    const ConstRec prec_k = (ConstRec)(sw.prec);

    // Filter one 32-sphere chunk (or a shorter tail, a multiple of 8): kCount sign bits, left-aligned
    // (bit 31 = first record of the chunk), 1 = rejected.
    auto filter_bits = [&](int base, auto count_tag) -> uint32_t {
        constexpr int kCount = decltype(count_tag)::value;
        uint32_t mask = 0u;
#pragma unroll
        for (int jj = 0; jj < kCount; ++jj) {
            float sgn;
            if (kPlane) {
                float fx, fy, fz;
                if (kRecLds) { const float4 f = s_prec[base + jj]; fx = f.x; fy = f.y; fz = f.z; }
                else { const f4v f = prec_k[base + jj]; fx = f.x; fy = f.y; fz = f.z; }
                const float pu = fmaf(fx, ux, fmaf(fy, uz, nou));
                sgn = fmaf(-pu, pu, fz);
            } else {
                const f4v f = frec_k[base + jj];
                const float b = fmaf(f.x, hx, fmaf(f.y, hy, fmaf(f.z, hz, nod)));
                const float w = fmaf(f.x, p2x, fmaf(f.y, p2y, fmaf(f.z, p2z, noo)));
                const float e = fmaf(b, b, w);
                sgn = e - f.w;
            }
            mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(sgn), 31);   // append sign bit
        }
        return mask;
    };

    T.mark(1);
    // Super-chunks of up to 16 mask words (512 spheres).  The filter only stores each word and notes
    // whether it has a survivor; afterwards one loop per lane walks its surviving bits in ascending
    // sphere order and runs the exact test of sphere.cpp:15-52 with the shrinking upper bound of
    // hittablelist.cpp:21-29 (unfused arithmetic).
    for (int sbase = 0; sbase < sw.n_pad; sbase += 32 * kSuperChunkWords) {
        uint32_t nz = 0u;          // bit (31 - w) set: word w of this super-chunk has a survivor
        int words = 0;
        int base = sbase;
        const int send = (sw.n_pad - sbase < 32 * kSuperChunkWords) ? sw.n_pad : sbase + 32 * kSuperChunkWords;
        for (; base + 32 <= send; base += 32, ++words) {
            const uint32_t cand = ~filter_bits(base, std::integral_constant<int, 32>());
            mask_words[words * 64] = cand;
            nz = __builtin_amdgcn_alignbit(nz, cand | (0u - cand), 31);
        }
        if (base < send) {         // tail: 8, 16 or 24 records
            uint32_t m = 0u; int got = 0;
            for (; base < send; base += 8, got += 8) m = (m << 8) | filter_bits(base, std::integral_constant<int, 8>());
            const uint32_t cand = (~m) << (32 - got);
            mask_words[words * 64] = cand;
            nz = __builtin_amdgcn_alignbit(nz, cand | (0u - cand), 31);
            ++words;
        }
        nz <<= (32 - words);
        T.mark(1);
        // Pass A: exact discriminant of every survivor of the filter (sphere.cpp:17-25); the bits of
        // those with desc <= 0 are cleared in place.  Pass B: roots, range test and running minimum
        // (sphere.cpp:27-50, hittablelist.cpp:21-29) for what is left, in ascending index order.
        // Splitting keeps the square root and the two divisions out of the long first loop, where
        // some lane of the wave would otherwise need them on every trip.
        // Every word flagged in nz is non-zero, and padding records never pass the filter, so each bit
        // visited names a real sphere (k < n).
        uint32_t nzb = 0u;
        if (nz != 0u) {
            int w = __builtin_clz(nz);
            nz &= ~(0x80000000u >> w);
            uint32_t cand = mask_words[w * 64], keep = 0u;
            int cbase = sbase + (w << 5);
            for (;;) {
                const int jj = __builtin_clz(cand);
                const uint32_t bit = 0x80000000u >> jj;
                cand &= ~bit;
                const int k = cbase + jj;
                const float4 g = kLds ? s_geom[k] : sw.geom[k];
                const V3 oc = ray.o - mk(g.x, g.y, g.z);
                const float b = dot(oc, ray.d);
                const float cc = dot(oc, oc) - g.w;
                const float desc = b * b - qa * cc;
                if (desc > 0.0f) keep |= bit;
                if (cand == 0u) {
                    mask_words[w * 64] = keep;
                    nzb |= keep ? 0x80000000u >> w : 0u;
                    if (nz == 0u) break;
                    w = __builtin_clz(nz);
                    nz &= ~(0x80000000u >> w);
                    cand = mask_words[w * 64];
                    keep = 0u;
                    cbase = sbase + (w << 5);
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
                cand &= ~(0x80000000u >> jj);
                const int k = cbase + jj;
                const float4 g = kLds ? s_geom[k] : sw.geom[k];
                const V3 oc = ray.o - mk(g.x, g.y, g.z);
                const float b = dot(oc, ray.d);
                const float cc = dot(oc, oc) - g.w;
                const float desc = b * b - qa * cc;
                const float sq = sqrtf(desc);
                float temp = (-b - sq) / qa;
                if (!(temp < closest && temp > tmin)) temp = (-b + sq) / qa;
                if (temp < closest && temp > tmin) { closest = temp; hit = k; }
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
constexpr int kFewPaths = RT_FEW_PATHS;     // 2, 4 and 8 measure alike on the 1200x800 final scene at 10 spp, 16 and 24 are 4-6 % slower (tools/ab_few.sh)
template <bool kLds>
__device__ __forceinline__ void closest_hit_few(const RayD& ray, float tmin, const SweepScene& sw, const float4* s_geom,
                                                unsigned long long live, int lane, float& closest, int& hit)
{
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
        for (int base = 0; base < sw.n; base += 64) {
            const int k = base + lane;
            float b = 0.0f, desc = -1.0f;
            if (k < sw.n) {
                const float4 g = kLds ? s_geom[k] : sw.geom[k];
                const V3 oc = r.o - mk(g.x, g.y, g.z);
                b = dot(oc, r.d);
                const float cc = dot(oc, oc) - g.w;
                desc = b * b - qa * cc;
            }
            if (__any(desc > 0.0f)) {
                if (desc > 0.0f) {
                    const float sq = sqrtf(desc);
                    float temp = (-b - sq) / qa;
                    if (!(temp < best && temp > tmin)) temp = (-b + sq) / qa;
                    if (temp < best && temp > tmin) { best = temp; best_k = k; }
                }
            }
        }
        // wave minimum of (t, index): smaller t wins, equal t goes to the lower index
        for (int off = 1; off < 64; off <<= 1) {
            const float ot = __shfl_xor(best, off); const int ok = __shfl_xor(best_k, off);
            if (ot < best || (ot == best && (unsigned)ok < (unsigned)best_k)) { best = ot; best_k = ok; }
        }
        if (lane == src) { closest = best; hit = best_k; }
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
__device__ __forceinline__ void bvh_test_sphere(const RayD& ray, float qa, float tmin, const BvhScene& bv, const float4* s_geom, int k, float& closest, int& hit)
{
    const float4 g = kLds ? s_geom[k] : bv.geom[k];
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
__device__ __forceinline__ void closest_hit_bvh(const RayD& ray, float tmin, const BvhScene& bv, const float4* s_geom, uint32_t* pending, float& closest, int& hit)
{
    const float qa = dot(ray.d, ray.d);
    for (int j = 0; j < bv.n_big; ++j) bvh_test_sphere<kLds>(ray, qa, tmin, bv, s_geom, (int)bv.big[j], closest, hit);
    const float ix = 1.0f / ray.d.x, iy = 1.0f / ray.d.y, iz = 1.0f / ray.d.z;
    const float omag = 1e-6f * (1.0f + fabsf(ray.o.x) + fabsf(ray.o.y) + fabsf(ray.o.z));     // the ray's share of the absolute margin
    const float t_low = tmin < 0.0f ? -FLT_MAX : 0.0f;
    auto test_leaf = [&](int link) {
        const int code = -(link + 1), first = code >> 3, count = (code & 7) + 1;
        for (int q = 0; q < count; ++q) bvh_test_sphere<kLds>(ray, qa, tmin, bv, s_geom, (int)bv.leaf_index[first + q], closest, hit);
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
// the lane's stack (element k at stack[k * sstride]).
template <bool kLds>
__device__ __forceinline__ bool shade_step(RayD& ray, int& depth, float closest, int hit, uint32_t pix, uint32_t smp, int max_depth,
                                           KeyedUniforms& U, const float4* s_geom, const float4* s_mat, const Aux* s_aux,
                                           const float4* g_geom, const float4* g_mat, const Aux* g_aux,
                                           uint16_t* stack, size_t sstride, V3& sample, SecTimer& T)
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
                stack[(size_t)depth * sstride] = (uint16_t)hit; ++depth; ray = out; done = false;
            }
        }
        T.mark(6);
    } else {
        sample = sky(ray.d);
        for (int k = depth - 1; k >= 0; --k) {
            const int h = stack[(size_t)k * sstride];
            const float4 m = kLds ? s_mat[h] : g_mat[h];
            sample = mk(m.x, m.y, m.z) * sample;
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
// that the order in which tiles are WORKED on can differ from where their samples are STORED.  `store` is the
// item number in image-tile order, i.e. the slot of the sample buffer.
struct ItemGeometry { int width, height, band_rows, band_first, band_step, owned_rows, tiles_x; uint32_t Q; int s_begin, s_count; const uint32_t* tile_order; };
__device__ __forceinline__ bool decode_item(const ItemGeometry& ig, uint32_t item, uint32_t& pix, uint32_t& smp, int& i, int& j, uint32_t& store)
{
    const uint32_t ts = item >> 6, l = item & 63u;
    const uint32_t work_tile = ts / (uint32_t)ig.s_count, s_rel = ts - work_tile * (uint32_t)ig.s_count;
    const uint32_t tile = ig.tile_order ? ig.tile_order[work_tile] : work_tile;
    store = ((tile * (uint32_t)ig.s_count + s_rel) << 6) + l;
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

// ============================================================================ work order of the tiles
// A persistent wave keeps all its lanes busy only while there are items left to hand out; once the counter
// runs dry every wave drains the paths it still holds, and a wave with one 50-bounce path left costs as many
// issue slots per bounce as a full one.  What is in flight at that moment is whatever was started LAST, so
// tiles are worked on in order of how long their paths can get: tiles that show a metal or glass sphere
// first, tiles that show only diffuse spheres next, tiles where every probe ray leaves the scene (one ray per
// sample) last.  The order changes scheduling only: a sample's value and its slot in the sample buffer
// depend on the image tile, not on when it is traced.
struct TileArgs {
    const float4* geom; const Aux* aux; int n;
    CameraD cam; ItemGeometry ig; float tmin;
    uint32_t tiles; uint32_t* cls; uint32_t* order;
};

// Eight probe rays per tile (corners and edge midpoints of the tile's pixel rectangle, through the lens
// centre), one lane each; class = the largest over the probes of {0 miss, 1 lambertian, 2 metal/dielectric}.
__global__ void __launch_bounds__(256) k_tile_class(const TileArgs a)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    const uint32_t tile = t >> 3, sub = t & 7u;
    uint32_t cls = 0;
    if (tile < a.tiles) {
        const int ty = (int)(tile / (uint32_t)a.ig.tiles_x), tx = (int)(tile - (uint32_t)ty * (uint32_t)a.ig.tiles_x);
        const int r0 = ty * 8, r1 = (r0 + 7 < a.ig.owned_rows) ? r0 + 7 : a.ig.owned_rows - 1;
        const int y0 = owned_row_to_y2(a.ig.band_rows, a.ig.band_first, a.ig.band_step, r0);
        const int y1 = owned_row_to_y2(a.ig.band_rows, a.ig.band_first, a.ig.band_step, r1);
        const float x_lo = (float)(tx * 8), x_hi = (float)((tx * 8 + 8 < a.ig.width) ? tx * 8 + 8 : a.ig.width);
        const float v_hi = (float)(a.ig.height - y0), v_lo = (float)(a.ig.height - 1 - y1);
        // probes 0..7 on a 3 x 3 grid without its centre
        const int g = (int)(sub < 4u ? sub : sub + 1u);
        const int gx = g % 3, gy = g / 3;
        const float px = gx == 0 ? x_lo : gx == 1 ? 0.5f * (x_lo + x_hi) : x_hi;
        const float py = gy == 0 ? v_lo : gy == 1 ? 0.5f * (v_lo + v_hi) : v_hi;
        const RayD ray = get_ray(a.cam, px / (float)a.ig.width, py / (float)a.ig.height, 0.0f, 0.0f);
        float closest = FLT_MAX; int hit = -1;
        const float qa = dot(ray.d, ray.d);
        for (int k = 0; k < a.n; ++k) {
            const float4 gk = a.geom[k];
            const V3 oc = ray.o - mk(gk.x, gk.y, gk.z);
            const float b = dot(oc, ray.d);
            const float desc = b * b - qa * (dot(oc, oc) - gk.w);
            if (desc > 0.0f) {
                const float sq = sqrtf(desc);
                float temp = (-b - sq) / qa;
                if (!(temp < closest && temp > a.tmin)) temp = (-b + sq) / qa;
                if (temp < closest && temp > a.tmin) { closest = temp; hit = k; }
            }
        }
        if (hit >= 0) cls = a.aux[hit].type == RT_LAMBERTIAN ? 1u : 2u;
    }
    for (int off = 1; off < 8; off <<= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)cls, off); cls = o > cls ? o : cls; }
    if (tile < a.tiles && sub == 0u) a.cls[tile] = cls;
}

// Stable counting sort of the tiles by class, highest class first (one block).
__global__ void __launch_bounds__(1024) k_tile_order(const TileArgs a)
{
    __shared__ uint32_t s_count[3];
    __shared__ uint32_t s_wave[16][3];
    __shared__ uint32_t s_run[3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 3) s_count[threadIdx.x] = 0;
    __syncthreads();
    uint32_t mine[3] = { 0, 0, 0 };
    for (uint32_t i = threadIdx.x; i < a.tiles; i += 1024u) mine[a.cls[i]] += 1u;
    for (int c = 0; c < 3; ++c) if (mine[c]) atomicAdd(&s_count[c], mine[c]);
    __syncthreads();
    if (threadIdx.x == 0) { s_run[2] = 0; s_run[1] = s_count[2]; s_run[0] = s_count[2] + s_count[1]; }
    __syncthreads();
    for (uint32_t base = 0; base < a.tiles; base += 1024u) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t c = i < a.tiles ? a.cls[i] : 3u;
        uint32_t rank = 0;
        for (uint32_t k = 0; k < 3; ++k) {
            const unsigned long long m = __ballot(c == k);
            if (c == k) rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (lane == 0) s_wave[wave][k] = (uint32_t)__popcll(m);
        }
        __syncthreads();
        if (c < 3u) {
            uint32_t before = s_run[c];
            for (int w = 0; w < wave; ++w) before += s_wave[w][c];
            a.order[before + rank] = i;
        }
        __syncthreads();
        if (threadIdx.x < 3) { uint32_t tot = 0; for (int w = 0; w < 16; ++w) tot += s_wave[w][threadIdx.x]; s_run[threadIdx.x] += tot; }
        __syncthreads();
    }
}

template <bool kLds, bool kPlane, bool kRecLds = false, bool kBvh = false>
__global__ void __launch_bounds__(kTraceBlock) k_trace(const TraceArgs a)
{
    extern __shared__ float4 smem[];
    // LDS: [geom n][mat n][aux n (8 B)] when kLds.
    float4* s_geom = smem;
    float4* s_mat = smem + (kLds ? a.n : 0);
    Aux* s_aux = reinterpret_cast<Aux*>(smem + (kLds ? 2 * a.n : 0));
    if (kLds) {
        for (int i = threadIdx.x; i < a.n; i += kTraceBlock) { s_geom[i] = a.geom[i]; s_mat[i] = a.mat[i]; s_aux[i] = a.aux[i]; }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // Per-lane filter masks of one super-chunk: word w of lane l at mask_words[w * 64] (lanes contiguous).
    char* lds_after_scene = reinterpret_cast<char*>(smem) + (kLds ? (size_t)a.n * kLdsBytesPerSphere : 0);
    // (In RT_MODE_BVH the same area holds each lane's stack of pending nodes instead.)
    constexpr int kWordsPerLane = kBvh ? kBvhStack : kSuperChunkWords;
    uint32_t* mask_words = reinterpret_cast<uint32_t*>(lds_after_scene) + (size_t)wave * kWordsPerLane * 64 + lane;
    // Measurement variant (kRecLds): plane-filter records mirrored in LDS and read by broadcast ds_read_b128
    // instead of by scalar load.
    float4* s_prec = reinterpret_cast<float4*>(lds_after_scene + (size_t)(kTraceBlock / 64) * kWordsPerLane * 64 * sizeof(uint32_t));
    if (kRecLds) { for (int i = threadIdx.x; i < a.n_pad; i += kTraceBlock) s_prec[i] = a.prec[i]; }
    __syncthreads();
    // The recursion's attenuation chain is replayed innermost-first when a path reaches the sky, so
    // each lane keeps the indices of the spheres it bounced off.  One 2-byte store per bounce and a
    // short read-back per finished path: kept in HBM so that LDS does not limit occupancy (packing the
    // first levels into registers was measured 1.5 % slower: the loads are hidden by the other waves).
    uint16_t* stack = a.stacks + (size_t)blockIdx.x * kTraceBlock + threadIdx.x;
    const size_t sstride = a.stack_stride;

    KeyedUniforms U; U.k0 = a.k0; U.k1 = a.k1; U.slot(0, 0, 0); U.cur = U4{ 0, 0, 0, 0 };
    RayD ray; ray.o = mk(0, 0, 0); ray.d = mk(0, 0, 1);
    int depth = 0; uint32_t item = 0, pix = 0, smp = 0;
    bool active = false;
    uint32_t rays = 0, misses = 0;
    uint32_t pool_next = 0, pool_end = 0, pool_base = 0;      // wave-uniform
    float* staged = a.staging + ((size_t)blockIdx.x * (kTraceBlock / 64) + wave) * (kStagingRows * kChunkItems);
    bool exhausted = false;                    // wave-uniform
    bool first_chunk = true;                   // wave-uniform
    const float fnx = (float)a.width, fny = (float)a.height;
    SecTimer T; T.start();
    ItemGeometry ig; ig.width = a.width; ig.height = a.height; ig.band_rows = a.band_rows; ig.band_first = a.band_first; ig.band_step = a.band_step;
    ig.owned_rows = a.owned_rows; ig.tiles_x = a.tiles_x; ig.Q = a.Q; ig.s_begin = a.s_begin; ig.s_count = a.s_count; ig.tile_order = a.tile_order;
    SweepScene sw; sw.frec = a.frec; sw.prec = a.prec; sw.plane_s0 = a.plane_s0; sw.plane_rmin = a.plane_rmin; sw.geom = a.geom; sw.n = a.n; sw.n_pad = a.n_pad;

#ifdef RT_WAVE_TIMELINE
    const unsigned long long t_enter = __builtin_amdgcn_s_memrealtime(); unsigned long long t_dry = 0, trips_dry = 0, sparse_trips = 0, sparse_ticks = 0, t_trip = 0, t_first_sweep = 0, t_first_done = 0; bool was_sparse = false;
#endif
    for (;;) {
#ifdef RT_WAVE_TIMELINE
        if (exhausted) { if (t_dry == 0) t_dry = __builtin_amdgcn_s_memrealtime(); ++trips_dry; }
        { const unsigned long long now = __builtin_amdgcn_s_memrealtime(); if (was_sparse) { ++sparse_trips; sparse_ticks += now - t_trip; } t_trip = now; was_sparse = false; }
#endif
        // ---------------------------------------------------------------- hand out new items
        unsigned long long need = __ballot(!active);
        while (need != 0ull && !(exhausted && pool_next == pool_end)) {
            if (pool_next == pool_end) {
                unsigned int g = 0;
                // Chunks shrink towards the end of the work (down to a floor that keeps the hand-out cheap next to
                // the sweep) so that the waves run out of items at nearly the same time: with full-size chunks to the
                // last, a wave can start a whole chunk
                // (some twenty trips) just before the counter runs dry while its neighbours go idle.  What is left
                // is estimated from this wave's previous reservation and one chunk per resident wave since.
                uint32_t take = (uint32_t)a.chunk_items;
                {
                    const uint32_t seen = pool_end + a.resident_waves * take;
                    const uint32_t left = a.total_items > seen ? a.total_items - seen : 0u;
                    const uint32_t fair = (left / (a.resident_waves * 2u)) & ~63u;
                    if (fair < take) take = fair < a.min_chunk_items ? a.min_chunk_items : fair;
                }
                if (first_chunk) {
                    // every wave's first chunk is fixed by its position in the grid (the host starts the counter
                    // behind them): thousands of waves asking one counter word at the same instant would queue
                    // for some 70 us at the start of every launch
                    first_chunk = false;
                    take = (uint32_t)a.chunk_items;
                    g = (blockIdx.x * (uint32_t)(kTraceBlock / 64) + (uint32_t)wave) * take;
                } else {
                    if (lane == 0) g = atomicAdd(a.work_counter, take);
                    g = __builtin_amdgcn_readfirstlane(g);
                }
                if (g >= a.total_items) { exhausted = true; break; }
                pool_next = g; pool_base = g;
                pool_end = (a.total_items - g < take) ? a.total_items : g + take;
                // Generate the camera rays of the whole chunk now, all lanes busy, into this wave's staging
                // rows in HBM; handing an item to a lane later is then eight loads.  (Generating on demand
                // ran this code on every trip of the main loop for the third of the lanes that had finished.)
                for (uint32_t it = g + (uint32_t)lane; it < pool_end; it += 64u) {
                    uint32_t gp = 0xffffffffu, gs = 0u, gstore = 0u; int i = 0, j = 0;
                    RayD gr; gr.o = mk(0, 0, 0); gr.d = mk(0, 0, 0);
                    if (decode_item(ig, it, gp, gs, i, j, gstore)) gr = camera_ray(a.cam, gp, gs, i, j, fnx, fny, a.k0, a.k1);
                    else gp = 0xffffffffu;
                    const uint32_t slot = it - g;
                    staged[0 * kChunkItems + slot] = gr.o.x; staged[1 * kChunkItems + slot] = gr.o.y; staged[2 * kChunkItems + slot] = gr.o.z;
                    staged[3 * kChunkItems + slot] = gr.d.x; staged[4 * kChunkItems + slot] = gr.d.y; staged[5 * kChunkItems + slot] = gr.d.z;
                    staged[6 * kChunkItems + slot] = __uint_as_float(gp); staged[7 * kChunkItems + slot] = __uint_as_float(gs);
                    staged[8 * kChunkItems + slot] = __uint_as_float(gstore);
                }
                // the rows are read back by other lanes of this wave: drain the stores first
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            const uint32_t avail = pool_end - pool_next;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
            const uint32_t want = (uint32_t)__popcll(need);
            if (!active && rank < avail) {
                const uint32_t slot = pool_next + rank - pool_base;
                // device-scope loads: the staging rows are rewritten for every chunk, so a stale L1 line
                // from the previous chunk must not be served
                const uint32_t gp = __float_as_uint(staged_load(staged + 6 * kChunkItems + slot));
                if (gp != 0xffffffffu) {            // 0xffffffff marks a padding pixel of an edge tile
                    pix = gp;
                    smp = __float_as_uint(staged_load(staged + 7 * kChunkItems + slot));
                    item = __float_as_uint(staged_load(staged + 8 * kChunkItems + slot));      // where the sample is stored
                    ray.o = mk(staged_load(staged + 0 * kChunkItems + slot), staged_load(staged + 1 * kChunkItems + slot), staged_load(staged + 2 * kChunkItems + slot));
                    ray.d = mk(staged_load(staged + 3 * kChunkItems + slot), staged_load(staged + 4 * kChunkItems + slot), staged_load(staged + 5 * kChunkItems + slot));
                    depth = 0;
                    active = true;
                }
            }
            pool_next += (want < avail ? want : avail);
            need = __ballot(!active);
            // padding pixels leave their lane idle; they are re-served next trip, which is harmless
            if (want <= avail) break;
        }
        T.mark(0);
        if (!__any(active)) {
            if (exhausted && pool_next == pool_end) break;
            continue;
        }

        // A path of L bounces needs L trips of its wave, and with six waves taking turns on a SIMD a trip lasts
        // six times its issue time: a 50-bounce path started late keeps the whole launch waiting.  Waves that
        // hold a deep path are therefore given issue priority over their neighbours (same total work, the long
        // chains just finish sooner).
        {
            const int deepest = __any(active && depth >= RT_PRIO_3) ? 3 : __any(active && depth >= RT_PRIO_2) ? 2 : __any(active && depth >= RT_PRIO_1) ? 1 : 0;
            if (deepest == 3) __builtin_amdgcn_s_setprio(3);
            else if (deepest == 2) __builtin_amdgcn_s_setprio(2);
            else if (deepest == 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        float closest = FLT_MAX; int hit = -1;
        const unsigned long long live = __ballot(active);
#ifdef RT_WAVE_TIMELINE
        was_sparse = __popcll(live) <= kFewPaths && live != 0ull;
        if (t_first_sweep == 0) t_first_sweep = __builtin_amdgcn_s_memrealtime();
        else if (t_first_done == 0) t_first_done = __builtin_amdgcn_s_memrealtime();
#endif
        if (kBvh) { if (active) closest_hit_bvh<kLds>(ray, a.tmin, a.bvh, s_geom, mask_words, closest, hit); }
        else if (__popcll(live) <= kFewPaths) closest_hit_few<kLds>(ray, a.tmin, sw, s_geom, live, lane, closest, hit);
        else if (active) closest_hit<kLds, kPlane, kRecLds>(ray, a.tmin, sw, s_geom, s_prec, mask_words, closest, hit, T);
        if (active) {
            T.mark(4);
            ++rays; misses += hit < 0 ? 1u : 0u;
            // ------------------------------------------------------------ shade (main.cpp:13-39)
            V3 sample;
            const bool done = shade_step<kLds>(ray, depth, closest, hit, pix, smp, a.max_depth, U, s_geom, s_mat, s_aux, a.geom, a.mat, a.aux, stack, sstride, sample, T);
            if (done) {
                float* o = a.samples + (size_t)item * 3;
                o[0] = sample.x; o[1] = sample.y; o[2] = sample.z;
                active = false;
            }
            T.mark(7);
        }
    }
    T.flush(lane);
#ifdef RT_WAVE_TIMELINE
    {
        const uint32_t w = blockIdx.x * (kTraceBlock / 64) + wave;
        if (lane == 0 && w < 8192u) { g_wave_times[0][w] = t_enter; const unsigned long long t_exit = __builtin_amdgcn_s_memrealtime(); g_wave_times[1][w] = t_dry ? t_dry : t_exit; g_wave_times[2][w] = t_exit; g_wave_times[3][w] = trips_dry; g_wave_times[4][w] = sparse_trips; g_wave_times[5][w] = sparse_ticks; g_wave_times[6][w] = t_first_sweep; g_wave_times[7][w] = t_first_done; }
    }
#endif
    // one atomic per wave
    for (int off = 32; off > 0; off >>= 1) { rays += __shfl_down(rays, off); misses += __shfl_down(misses, off); }
    if (lane == 0 && rays) { atomicAdd(a.ray_counter, (unsigned long long)rays); atomicAdd(a.ray_counter + 1, (unsigned long long)misses); }
}

// Ordered per-pixel sum of a sample batch: col += sample for s ascending, exactly the
// accumulation order of main.cpp:124-134; the last batch applies `col /= ns` and quantises.
struct ResolveArgs {
    const float* samples; float* accum;     // accum: [Q][3] running sums
    uint32_t Q; int tiles_x; int s_count; int first, last;
    int width, owned_rows, spp;
    float* rgb_mean; int32_t* irgb;
};
__global__ void __launch_bounds__(256) k_resolve(const ResolveArgs a)
{
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    if (q >= a.Q) return;
    V3 col = a.first ? mk(0.0f, 0.0f, 0.0f) : mk(a.accum[3 * (size_t)q], a.accum[3 * (size_t)q + 1], a.accum[3 * (size_t)q + 2]);
    const uint32_t tile = q >> 6, l = q & 63u;
    const int ty = (int)(tile / (uint32_t)a.tiles_x), tx = (int)(tile - (uint32_t)ty * (uint32_t)a.tiles_x);
    const int i = tx * 8 + (int)(l & 7u), r = ty * 8 + (int)(l >> 3);
    if (i >= a.width || r >= a.owned_rows) return;
    const float* sp = a.samples + 3 * ((size_t)tile * a.s_count * 64 + l);       // item = (tile * s_count + s) * 64 + l
    for (int s = 0; s < a.s_count; ++s, sp += 3 * 64) col = col + mk(sp[0], sp[1], sp[2]);
    if (!a.last) { a.accum[3 * (size_t)q] = col.x; a.accum[3 * (size_t)q + 1] = col.y; a.accum[3 * (size_t)q + 2] = col.z; return; }
    const float inv = 1.0f / (float)a.spp;          // Vec3::operator/=(float), vec3.cpp:109-118
    const float mr = col.x * inv, mg = col.y * inv, mb = col.z * inv;
    const size_t o = ((size_t)r * a.width + i) * 3;
    if (a.rgb_mean) { a.rgb_mean[o] = mr; a.rgb_mean[o + 1] = mg; a.rgb_mean[o + 2] = mb; }
    if (a.irgb) { a.irgb[o] = quantise(mr); a.irgb[o + 1] = quantise(mg); a.irgb[o + 2] = quantise(mb); }
}

// ============================================================================ wavefront mode
// The same path tracer as three kernels over ray queues in HBM (mode RT_MODE_WAVEFRONT).  A pool of P path
// slots holds one ray each; a queue lists the live slots.  Per bounce: k_wf_intersect runs the closest-hit
// sweep for every queued ray (all lanes busy, nothing else in the kernel); k_wf_shade scatters or finishes
// each ray, gives a finished slot the next (pixel, sample) item, and writes the surviving slot ids to the
// next queue compacted with a wave ballot + prefix count and one atomic per wave.
struct WfArgs {
    SweepScene sw; const float4* mat; const Aux* aux;
    CameraD cam; ItemGeometry ig;
    int max_depth; float tmin; uint32_t k0, k1;
    uint32_t total_items, P;
    float* ray;                 // [6][P]  o.xyz, d.xyz
    uint32_t* pixsmp;           // [3][P]  pixel key, sample, item
    int* depth;                 // [P]
    float* hit_t; int* hit_idx; // [P]
    uint16_t* stacks;           // [RT_MAX_DEPTH][P]
    const uint32_t* q_in; uint32_t* q_out; const uint32_t* n_in; uint32_t* n_out;
    unsigned int* work_counter; unsigned long long* ray_counter;
    float* samples;
};

__global__ void __launch_bounds__(256) k_wf_init(const WfArgs a)
{
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    const uint32_t first = a.P < a.total_items ? a.P : a.total_items;
    if (s == 0) { *a.work_counter = first; }
    if (s >= first) return;
    uint32_t pix = 0, smp = 0; int i = 0, j = 0;
    uint32_t store = 0;
    const bool ok = decode_item(a.ig, s, pix, smp, i, j, store);
    // every slot starts live; a padding pixel becomes a ray that is retired (and replaced) by the first shade pass
    RayD r; r.o = mk(0, 0, 0); r.d = mk(0, 0, 0);
    if (ok) r = camera_ray(a.cam, pix, smp, i, j, (float)a.ig.width, (float)a.ig.height, a.k0, a.k1);
    a.ray[0 * (size_t)a.P + s] = r.o.x; a.ray[1 * (size_t)a.P + s] = r.o.y; a.ray[2 * (size_t)a.P + s] = r.o.z;
    a.ray[3 * (size_t)a.P + s] = r.d.x; a.ray[4 * (size_t)a.P + s] = r.d.y; a.ray[5 * (size_t)a.P + s] = r.d.z;
    a.pixsmp[0 * (size_t)a.P + s] = ok ? pix : 0xffffffffu; a.pixsmp[1 * (size_t)a.P + s] = smp; a.pixsmp[2 * (size_t)a.P + s] = s;
    a.depth[s] = 0;
    a.q_out[s] = s;
    if (s == 0) *a.n_out = first;
}

template <bool kLds, bool kPlane>
__global__ void __launch_bounds__(kTraceBlock) k_wf_intersect(const WfArgs a)
{
    extern __shared__ float4 smem[];
    float4* s_geom = smem;
    if (kLds) { for (int i = threadIdx.x; i < a.sw.n; i += kTraceBlock) s_geom[i] = a.sw.geom[i]; }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t* mask_words = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(smem) + (kLds ? (size_t)a.sw.n * sizeof(float4) : 0)) + (size_t)wave * kSuperChunkWords * 64 + lane;
    __syncthreads();
    const uint32_t n_in = *a.n_in;
    SecTimer T; T.start();
    // grid-stride over the queue so that a block's LDS copy of the scene serves many rays
    for (uint32_t e = blockIdx.x * (uint32_t)kTraceBlock + threadIdx.x; e < n_in; e += gridDim.x * (uint32_t)kTraceBlock) {
        const uint32_t s = a.q_in[e];
        RayD ray;
        ray.o = mk(a.ray[0 * (size_t)a.P + s], a.ray[1 * (size_t)a.P + s], a.ray[2 * (size_t)a.P + s]);
        ray.d = mk(a.ray[3 * (size_t)a.P + s], a.ray[4 * (size_t)a.P + s], a.ray[5 * (size_t)a.P + s]);
        float closest = FLT_MAX; int hit = -1;
        if (a.pixsmp[s] != 0xffffffffu) closest_hit<kLds, kPlane, false>(ray, a.tmin, a.sw, s_geom, nullptr, mask_words, closest, hit, T);
        a.hit_t[s] = closest; a.hit_idx[s] = hit;
    }
}

// One reservation on a global counter for a whole block: every wave contributes `mine` (wave-uniform), thread 0 adds
// the block's total with a single atomic, and each wave gets the start of its share.  Returns the block's total.
// A single counter word takes about 88 atomics per microsecond on this chip; one per wave, with 131 072 waves per
// pass, made the counters the bottleneck of the shade pass.
constexpr int kWfShadeBlock = 1024;
__device__ __forceinline__ uint32_t block_reserve(uint32_t* counter, uint32_t mine, int wave, int lane, uint32_t* s_counts, uint32_t* s_base, uint32_t& my_base)
{
    if (lane == 0) s_counts[wave] = mine;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (int w = 0; w < kWfShadeBlock / 64; ++w) { const uint32_t c = s_counts[w]; before += w < wave ? c : 0u; total += c; }
    if (threadIdx.x == 0 && total != 0u) *s_base = atomicAdd(counter, total);
    __syncthreads();
    my_base = *s_base + before;
    __syncthreads();            // s_counts and s_base are reused by the next reservation
    return total;
}

__global__ void __launch_bounds__(kWfShadeBlock) k_wf_shade(const WfArgs a)
{
    __shared__ uint32_t s_counts[kWfShadeBlock / 64];
    __shared__ uint32_t s_base;
    __shared__ unsigned long long s_rays[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 2) s_rays[threadIdx.x] = 0ull;
    const uint32_t n_in = *a.n_in;
    const uint32_t e = blockIdx.x * (uint32_t)kWfShadeBlock + threadIdx.x;
    bool live = e < n_in;
    bool traced = false;          // this lane's ray went through a closest-hit query
    bool missed = false;          // ... and hit nothing
    bool need = false, fresh = false;
    V3 ray_o = mk(0, 0, 0), ray_d = mk(0, 0, 0); int depth_out = 0; uint32_t pix_out = 0, smp_out = 0;
    uint32_t s = 0;
    SecTimer T; T.start();
    if (live) {
        s = a.q_in[e];
        RayD ray;
        ray.o = mk(a.ray[0 * (size_t)a.P + s], a.ray[1 * (size_t)a.P + s], a.ray[2 * (size_t)a.P + s]);
        ray.d = mk(a.ray[3 * (size_t)a.P + s], a.ray[4 * (size_t)a.P + s], a.ray[5 * (size_t)a.P + s]);
        uint32_t pix = a.pixsmp[0 * (size_t)a.P + s], smp = a.pixsmp[1 * (size_t)a.P + s];
        const uint32_t item = a.pixsmp[2 * (size_t)a.P + s];
        int depth = a.depth[s];
        bool done = true; V3 sample = mk(0, 0, 0);
        KeyedUniforms U; U.k0 = a.k0; U.k1 = a.k1; U.slot(pix, smp, 0); U.cur = U4{ 0, 0, 0, 0 };
        if (pix != 0xffffffffu) {
            traced = true; missed = a.hit_idx[s] < 0;
            done = shade_step<false>(ray, depth, a.hit_t[s], a.hit_idx[s], pix, smp, a.max_depth, U, nullptr, nullptr, nullptr,
                                     a.sw.geom, a.mat, a.aux, a.stacks + s, (size_t)a.P, sample, T);
            if (done) { float* o = a.samples + (size_t)item * 3; o[0] = sample.x; o[1] = sample.y; o[2] = sample.z; }
        }
        need = done;
        if (done) live = false;
        ray_o = ray.o; ray_d = ray.d; depth_out = depth; pix_out = pix; smp_out = smp;
    }
    // A finished slot takes the next (pixel, sample) item.  Items are reserved per block: ballot of the lanes
    // that need one, the waves' counts summed in LDS, one atomic for the block, prefix rank per lane.  Padding
    // pixels of edge tiles are skipped by asking again (the loop count is the same for all threads of the block).
    uint32_t item_out = 0;
    for (;;) {
        const unsigned long long want = __ballot(need);
        uint32_t base = 0;
        if (block_reserve(a.work_counter, (uint32_t)__popcll(want), wave, lane, s_counts, &s_base, base) == 0u) break;
        if (need) {
            const uint32_t it = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(want >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)want, 0u));
            int i = 0, j = 0; uint32_t store_unused = 0;
            if (it >= a.total_items) { need = false; }
            else if (decode_item(a.ig, it, pix_out, smp_out, i, j, store_unused)) {
                const RayD r = camera_ray(a.cam, pix_out, smp_out, i, j, (float)a.ig.width, (float)a.ig.height, a.k0, a.k1);
                ray_o = r.o; ray_d = r.d; depth_out = 0; item_out = it; live = true; need = false; fresh = true;
            }
        }
    }
    if (live) {
        a.ray[0 * (size_t)a.P + s] = ray_o.x; a.ray[1 * (size_t)a.P + s] = ray_o.y; a.ray[2 * (size_t)a.P + s] = ray_o.z;
        a.ray[3 * (size_t)a.P + s] = ray_d.x; a.ray[4 * (size_t)a.P + s] = ray_d.y; a.ray[5 * (size_t)a.P + s] = ray_d.z;
        a.depth[s] = depth_out;
        if (fresh) { a.pixsmp[0 * (size_t)a.P + s] = pix_out; a.pixsmp[1 * (size_t)a.P + s] = smp_out; a.pixsmp[2 * (size_t)a.P + s] = item_out; }
    }
    const unsigned long long did = __ballot(traced), sky_end = __ballot(missed);
    if (lane == 0 && did != 0ull) { atomicAdd(&s_rays[0], (unsigned long long)__popcll(did)); atomicAdd(&s_rays[1], (unsigned long long)__popcll(sky_end)); }
    // compaction: ballot of surviving lanes, prefix rank within the wave, the waves' counts summed in LDS, one atomic per block
    const unsigned long long alive = __ballot(live);
    uint32_t base = 0;
    (void)block_reserve(a.n_out, (uint32_t)__popcll(alive), wave, lane, s_counts, &s_base, base);      // its barriers also publish s_rays
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(alive >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)alive, 0u));
    if (live) a.q_out[base + rank] = s;
    if (threadIdx.x == 0 && s_rays[0] != 0ull) { atomicAdd(a.ray_counter, s_rays[0]); atomicAdd(a.ray_counter + 1, s_rays[1]); }
}

// ============================================================================ host side
static thread_local std::string g_error;

int fail(rt_context* ctx, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (ctx) ctx->error = buf; else g_error = buf;
    return code;
}

#ifdef RT_SECTION_TIMERS
extern "C" int rt_debug_section_ticks(unsigned long long* out8, int reset)
{
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_section_ticks), 8 * sizeof(unsigned long long)) != hipSuccess) return -3;
    if (reset) { unsigned long long z[8] = { 0 }; if (hipMemcpyToSymbol(HIP_SYMBOL(g_section_ticks), z, sizeof z) != hipSuccess) return -3; }
    return 0;
}
#endif
#ifdef RT_WAVE_TIMELINE
extern "C" int rt_debug_wave_times(unsigned long long* out_6x8192)   /* 8 rows of 8192 */
{
    return hipMemcpyFromSymbol(out_6x8192, HIP_SYMBOL(g_wave_times), 8 * 8192 * sizeof(unsigned long long)) == hipSuccess ? 0 : -3;
}
#endif

extern "C" {

int rt_device_count(int* count)
{
    if (!count) return fail(nullptr, RT_ERR_INVALID, "rt_device_count: null output");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(nullptr, RT_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return RT_OK;
}

int rt_create(int device_ordinal, rt_context** out)
{
    if (!out) return fail(nullptr, RT_ERR_INVALID, "rt_create: null output");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(nullptr, RT_ERR_NO_DEVICE, "rt_create: no HIP device visible");
    if (device_ordinal < 0 || device_ordinal >= n) return fail(nullptr, RT_ERR_NO_DEVICE, "rt_create: device %d out of range [0,%d)", device_ordinal, n);
    rt_context* ctx = new (std::nothrow) rt_context;
    if (!ctx) return fail(nullptr, RT_ERR_HIP, "rt_create: out of host memory");
    ctx->device = device_ordinal;
    // RT_AMD_FILTER=disc selects the full-discriminant filter instead of the plane filter (A/B runs).
    { const char* f = getenv("RT_AMD_FILTER"); ctx->plane_filter = !(f && strcmp(f, "disc") == 0); }
    { const char* f = getenv("RT_AMD_RECORDS"); ctx->records_in_lds = (f && strcmp(f, "lds") == 0); }
    // RT_AMD_TILE_ORDER=image works through the tiles in image order instead of longest-paths-first (A/B runs).
    // RT_AMD_SAMPLE_BUFFER_KB caps the sample buffer (tests use it to run the multi-batch path on small images).
    { const char* f = getenv("RT_AMD_SAMPLE_BUFFER_KB"); const long long kb = f ? atoll(f) : 0; if (kb > 0) ctx->sample_buffer_cap = (uint64_t)kb << 10; }
    { const char* f = getenv("RT_AMD_TILE_ORDER"); ctx->tile_ordering = !(f && strcmp(f, "image") == 0); }
    hipError_t e = hipSetDevice(device_ordinal);
    if (e == hipSuccess) e = hipMalloc(&ctx->d_rays, 2 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc(&ctx->d_work, sizeof(unsigned int));
    if (e == hipSuccess) { hipDeviceProp_t prop; e = hipGetDeviceProperties(&prop, device_ordinal); if (e == hipSuccess) ctx->num_cu = prop.multiProcessorCount; }
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev0);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev1);
    if (e != hipSuccess) { fail(nullptr, RT_ERR_HIP, "rt_create: %s", hipGetErrorString(e)); rt_destroy(ctx); return RT_ERR_HIP; }
    *out = ctx;
    return RT_OK;
}

void rt_destroy(rt_context* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->pending) (void)hipStreamSynchronize(ctx->last_stream);
    (void)hipFree(ctx->d_geom); (void)hipFree(ctx->d_mat); (void)hipFree(ctx->d_aux); (void)hipFree(ctx->d_rays); (void)hipFree(ctx->d_ppm_scratch); (void)hipFree(ctx->d_out_rgb); (void)hipFree(ctx->d_out_irgb);
    (void)hipFree(ctx->d_bvh_nodes); (void)hipFree(ctx->d_bvh_leaf); (void)hipFree(ctx->d_bvh_big);
    (void)hipFree(ctx->d_frec); (void)hipFree(ctx->d_prec); (void)hipFree(ctx->d_samples); (void)hipFree(ctx->d_accum); (void)hipFree(ctx->d_work); (void)hipFree(ctx->d_stacks); (void)hipFree(ctx->d_staging); (void)hipFree(ctx->d_tile_cls); (void)hipFree(ctx->d_tile_order);
    (void)hipFree(ctx->wf_ray); (void)hipFree(ctx->wf_pixsmp); (void)hipFree(ctx->wf_depth); (void)hipFree(ctx->wf_hit_t); (void)hipFree(ctx->wf_hit_idx);
    (void)hipFree(ctx->wf_stacks); (void)hipFree(ctx->wf_q[0]); (void)hipFree(ctx->wf_q[1]); (void)hipFree(ctx->wf_n);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    delete ctx;
}

const char* rt_last_error(const rt_context* ctx) { return ctx ? ctx->error.c_str() : g_error.c_str(); }

int rt_upload_scene(rt_context* ctx, int n, const float* cx, const float* cy, const float* cz, const float* radius,
                    const int32_t* material, const float* albedo_rgb, const float* param)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "rt_upload_scene: null context");
    if (n <= 0 || !cx || !cy || !cz || !radius || !material || !albedo_rgb || !param) return fail(ctx, RT_ERR_INVALID, "rt_upload_scene: null array or n <= 0");
    if (n > 65535) return fail(ctx, RT_ERR_TOO_LARGE, "rt_upload_scene: %d spheres exceeds the 65535 limit of the 16-bit hit stack", n);
    std::vector<float4> geom(n), mat(n); std::vector<Aux> aux(n);
    const int n_pad = (n + 7) / 8 * 8;
    // Filter records (cx, cy, cz, k), k = |c|^2 - r^2 - slack_i.  With u = 2^-24, M = |o| + |c| and the
    // ray direction normalised, the filter's own rounding error is below 32 u M^2 and the reference's
    // discriminant (divided by |d|^2) is off by less than 32 u (|o - c|^2 + r^2); since
    // M^2 <= 2 (|o|^2 + |c|^2), slack_i = 160 u |c|^2 + 48 u r^2 here plus 160 u |o|^2 per ray in the
    // kernel makes "filter negative" imply "reference discriminant not positive".
    std::vector<float4> frec(n_pad, make_float4(0.0f, 0.0f, 0.0f, FLT_MAX));
    for (int i = 0; i < n; ++i) {
        const double cc = (double)cx[i] * cx[i] + (double)cy[i] * cy[i] + (double)cz[i] * cz[i], rr = (double)radius[i] * radius[i];
        const double k = cc - rr - 5.9604644775390625e-08 * (160.0 * cc + 48.0 * rr);
        float kf = (float)k;
        if ((double)kf > k) kf = nextafterf(kf, -INFINITY);
        frec[i] = make_float4(cx[i], cy[i], cz[i], kf);
    }
    // Plane-filter records (cx, cz, R_i = r^2 + slack_i, 0).  u = 2^-24.  Allowance for the reference's own
    // discriminant rounding: 32 u (|o-c|^2 + r^2) <= 64 u (|o|^2 + |c|^2) + 32 u r^2; this form's rounding
    // of pu (< 8 u (|c| + |o|)) moves pu^2 near the threshold by < 16 u |r| (|c| + |o|)
    // <= 8 u (2 r^2 + |c|^2 + |o|^2).  slack_i = u (128 |c|^2 + 64 r^2), S_ray = 128 u |o|^2 (kernel).
    std::vector<float4> prec(n_pad, make_float4(0.0f, 0.0f, -FLT_MAX, 0.0f));
    double rmin_R = 1e30;
    for (int i = 0; i < n; ++i) {
        const double rr = (double)radius[i] * radius[i];
        const double cc = (double)cx[i] * cx[i] + (double)cy[i] * cy[i] + (double)cz[i] * cz[i];
        const double R = rr + 5.9604644775390625e-08 * (128.0 * cc + 64.0 * rr) + 1e-30;
        float rf = (float)R; if ((double)rf < R) rf = nextafterf(rf, INFINITY);
        prec[i] = make_float4(cx[i], cz[i], rr > 1e-12 ? rf : FLT_MAX, 0.0f);     // zero-radius spheres always go to the exact test
        if (rr > 1e-12 && (double)rf < rmin_R) rmin_R = (double)rf;
    }
    const float plane_s0 = 1e-30f;
    if (rmin_R > 1e29) rmin_R = 1.0;
    const float plane_rmin = nextafterf((float)rmin_R, 0.0f);
    for (int i = 0; i < n; ++i) {
        if (material[i] < RT_LAMBERTIAN || material[i] > RT_DIELECTRIC) return fail(ctx, RT_ERR_INVALID, "rt_upload_scene: sphere %d has unknown material %d", i, material[i]);
        geom[i] = make_float4(cx[i], cy[i], cz[i], radius[i] * radius[i]);
        const bool glass = material[i] == RT_DIELECTRIC;   // Dielectric::Scatter sets attenuation (1,1,1)
        mat[i] = make_float4(glass ? 1.0f : albedo_rgb[3 * i], glass ? 1.0f : albedo_rgb[3 * i + 1], glass ? 1.0f : albedo_rgb[3 * i + 2], param[i]);
        aux[i].radius = radius[i]; aux[i].type = material[i];
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->pending) { HIP_TRY(ctx, hipStreamSynchronize(ctx->last_stream)); ctx->pending = false; }
    (void)hipFree(ctx->d_geom); (void)hipFree(ctx->d_mat); (void)hipFree(ctx->d_aux); (void)hipFree(ctx->d_frec); (void)hipFree(ctx->d_prec);
    ctx->d_geom = nullptr; ctx->d_mat = nullptr; ctx->d_aux = nullptr; ctx->d_frec = nullptr; ctx->d_prec = nullptr; ctx->n = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->d_prec, sizeof(float4) * n_pad));
    HIP_TRY(ctx, hipMemcpy(ctx->d_prec, prec.data(), sizeof(float4) * n_pad, hipMemcpyHostToDevice));
    ctx->plane_s0 = plane_s0; ctx->plane_rmin = plane_rmin;
    HIP_TRY(ctx, hipMalloc(&ctx->d_frec, sizeof(float4) * n_pad));
    HIP_TRY(ctx, hipMemcpy(ctx->d_frec, frec.data(), sizeof(float4) * n_pad, hipMemcpyHostToDevice));
    ctx->n_pad = n_pad;
    // n_pad entries: a ray with a NaN direction passes every filter record, padding included, and the exact pass
    // then reads geom[k] for k up to n_pad - 1 (the NaN discriminant rejects it)
    HIP_TRY(ctx, hipMalloc(&ctx->d_geom, sizeof(float4) * n_pad));
    HIP_TRY(ctx, hipMemset(ctx->d_geom, 0, sizeof(float4) * n_pad));
    HIP_TRY(ctx, hipMalloc(&ctx->d_mat, sizeof(float4) * n));
    HIP_TRY(ctx, hipMalloc(&ctx->d_aux, sizeof(Aux) * n));
    HIP_TRY(ctx, hipMemcpy(ctx->d_geom, geom.data(), sizeof(float4) * n, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->d_mat, mat.data(), sizeof(float4) * n, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->d_aux, aux.data(), sizeof(Aux) * n, hipMemcpyHostToDevice));
    {
        BvhHost tree;
        if (!build_bvh(n, cx, cy, cz, radius, &tree)) return fail(ctx, RT_ERR_TOO_LARGE, "rt_upload_scene: sphere hierarchy deeper than %d levels", kBvhStack - 2);
        (void)hipFree(ctx->d_bvh_nodes); (void)hipFree(ctx->d_bvh_leaf); (void)hipFree(ctx->d_bvh_big);
        ctx->d_bvh_nodes = nullptr; ctx->d_bvh_leaf = nullptr; ctx->d_bvh_big = nullptr;
        HIP_TRY(ctx, hipMalloc(&ctx->d_bvh_nodes, sizeof(float4) * tree.nodes.size()));
        HIP_TRY(ctx, hipMemcpy(ctx->d_bvh_nodes, tree.nodes.data(), sizeof(float4) * tree.nodes.size(), hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMalloc(&ctx->d_bvh_leaf, sizeof(uint32_t) * tree.leaf_index.size()));
        HIP_TRY(ctx, hipMemcpy(ctx->d_bvh_leaf, tree.leaf_index.data(), sizeof(uint32_t) * tree.leaf_index.size(), hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMalloc(&ctx->d_bvh_big, sizeof(uint32_t) * (tree.big.size() + 1)));
        if (!tree.big.empty()) HIP_TRY(ctx, hipMemcpy(ctx->d_bvh_big, tree.big.data(), sizeof(uint32_t) * tree.big.size(), hipMemcpyHostToDevice));
        ctx->bvh_n_big = (int)tree.big.size(); ctx->bvh_depth = tree.depth; ctx->bvh_nodes = tree.node_count();
    }
    ctx->n = n; ++ctx->scene_version;
    return RT_OK;
}

int rt_set_camera(rt_context* ctx, const rt_camera_pod* cam)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "rt_set_camera: null context");
    if (!cam) return fail(ctx, RT_ERR_INVALID, "rt_set_camera: null camera");
    ctx->cam = to_device(*cam);
    ctx->have_camera = true; ++ctx->camera_version;
    return RT_OK;
}

int64_t rt_owned_rows(const rt_params* p)
{
    if (!params_ok(p)) return 0;
    const int64_t nbands = ((int64_t)p->height + p->band_rows - 1) / p->band_rows;
    int64_t rows = 0;
    for (int64_t b = p->band_first; b < nbands; b += p->band_step) {
        const int64_t y0 = b * p->band_rows, y1 = y0 + p->band_rows;
        rows += (y1 < p->height ? y1 : p->height) - y0;
    }
    return rows;
}

int64_t rt_owned_pixels(const rt_params* p) { return rt_owned_rows(p) * (p ? p->width : 0); }

int rt_render_device(rt_context* ctx, const rt_params* p, float* d_rgb_mean, int32_t* d_irgb, void* stream)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "rt_render_device: null context");
    if (!params_ok(p)) return fail(ctx, RT_ERR_INVALID, "rt_render_device: invalid parameters");
    if (!ctx->n || !ctx->have_camera) return fail(ctx, RT_ERR_STATE, "rt_render_device: scene and camera must be set first");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int owned = (int)rt_owned_rows(p);
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_rays, 0, 2 * sizeof(unsigned long long), st));
    HIP_TRY(ctx, hipEventRecord(ctx->ev0, st));
    if (owned > 0 && p->mode == RT_MODE_PIXEL) {
        RenderArgs a;
        a.geom = ctx->d_geom; a.mat = ctx->d_mat; a.aux = ctx->d_aux; a.n = ctx->n; a.cam = ctx->cam;
        a.width = p->width; a.height = p->height; a.spp = p->spp; a.max_depth = p->max_depth; a.tmin = p->tmin;
        a.k0 = (uint32_t)p->seed; a.k1 = (uint32_t)(p->seed >> 32);
        a.band_rows = p->band_rows; a.band_first = p->band_first; a.band_step = p->band_step; a.owned_rows = owned;
        a.rgb_mean = d_rgb_mean; a.irgb = d_irgb; a.ray_counter = ctx->d_rays;
        const dim3 grid((p->width + 15) / 16, (owned + 15) / 16), block(kBlock);
        if (ctx->n <= kMaxLdsSpheres) hipLaunchKernelGGL(k_render_mega<true>, grid, block, (size_t)ctx->n * kLdsBytesPerSphere, st, a);
        else hipLaunchKernelGGL(k_render_mega<false>, grid, block, 0, st, a);
        HIP_TRY(ctx, hipGetLastError());
    } else if (owned > 0) {
        const int tiles_x = (p->width + 7) / 8, tiles_y = (owned + 7) / 8;
        const uint64_t Q = (uint64_t)tiles_x * tiles_y * 64;
        // sample batches: the per-batch sample buffer stays below 16 GiB (of 288 GB of HBM) and item ids below 2^31;
        // every batch ends with the drain of its longest paths, so fewer, larger batches are better
        uint64_t per_batch = ctx->sample_buffer_cap / (Q * 12);
        if (per_batch > (1ull << 31) / Q) per_batch = (1ull << 31) / Q;
        if (per_batch == 0) return fail(ctx, RT_ERR_TOO_LARGE, "rt_render_device: %llu pixels in one call exceed the sample-buffer limit; split the image into bands", (unsigned long long)Q);
        const int batches = (int)((p->spp + per_batch - 1) / per_batch);
        const int s_per = (p->spp + batches - 1) / batches;
        const size_t need = (size_t)Q * s_per * 12;
        if (ctx->samples_bytes < need) {
            HIP_TRY(ctx, hipStreamSynchronize(st));
            (void)hipFree(ctx->d_samples); ctx->d_samples = nullptr; ctx->samples_bytes = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->d_samples, need)); ctx->samples_bytes = need;
        }
        if (batches > 1 && ctx->accum_bytes < Q * 12) {
            HIP_TRY(ctx, hipStreamSynchronize(st));
            (void)hipFree(ctx->d_accum); ctx->d_accum = nullptr; ctx->accum_bytes = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->d_accum, Q * 12)); ctx->accum_bytes = Q * 12;
        }
        const bool lds = ctx->n <= kMaxLdsSpheres;
        const bool plane = ctx->plane_filter;
        const bool bvh = p->mode == RT_MODE_BVH;
        const bool reclds = !bvh && ctx->records_in_lds && lds && plane && ctx->n_pad <= 2048;
        const size_t shmem = (lds ? (size_t)ctx->n * kLdsBytesPerSphere : 0) + (size_t)(kTraceBlock / 64) * (bvh ? kBvhStack : kSuperChunkWords) * 64 * sizeof(uint32_t) + (reclds ? (size_t)ctx->n_pad * sizeof(float4) : 0);
        int& bpc = ctx->trace_blocks_per_cu[bvh ? (lds ? 6 : 5) : reclds ? 4 : (lds ? 1 : 0) + (plane ? 2 : 0)];
        if (bpc == 0) {
            int nb = 0;
            auto kern = bvh ? (lds ? k_trace<true, true, false, true> : k_trace<false, true, false, true>)
                      : reclds ? k_trace<true, true, true> : lds ? (plane ? k_trace<true, true> : k_trace<true, false>) : (plane ? k_trace<false, true> : k_trace<false, false>);
            hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, kTraceBlock, shmem);
            if (e != hipSuccess || nb <= 0) nb = 1;
            bpc = nb;
        }
        const size_t lanes = (size_t)ctx->num_cu * bpc * kTraceBlock;
        if (ctx->stacks_lanes < lanes) {
            HIP_TRY(ctx, hipStreamSynchronize(st));
            (void)hipFree(ctx->d_stacks); ctx->d_stacks = nullptr; ctx->stacks_lanes = 0;
            (void)hipFree(ctx->d_staging); ctx->d_staging = nullptr;
            HIP_TRY(ctx, hipMalloc(&ctx->d_stacks, lanes * RT_MAX_DEPTH * sizeof(uint16_t)));
            HIP_TRY(ctx, hipMalloc(&ctx->d_staging, (lanes / 64) * kStagingRows * kChunkItems * sizeof(float)));
            ctx->stacks_lanes = lanes;
        }
        const uint32_t n_tiles = (uint32_t)(Q / 64);
        const bool ordered = ctx->tile_ordering && (p->mode == RT_MODE_MEGAKERNEL || bvh);
        if (ordered) {
            if (ctx->tile_cap < n_tiles) {
                HIP_TRY(ctx, hipStreamSynchronize(st));
                (void)hipFree(ctx->d_tile_cls); (void)hipFree(ctx->d_tile_order); ctx->d_tile_cls = ctx->d_tile_order = nullptr; ctx->tile_cap = 0; ctx->tile_key_valid = false;
                HIP_TRY(ctx, hipMalloc(&ctx->d_tile_cls, (size_t)n_tiles * sizeof(uint32_t)));
                HIP_TRY(ctx, hipMalloc(&ctx->d_tile_order, (size_t)n_tiles * sizeof(uint32_t)));
                ctx->tile_cap = n_tiles;
            }
            const rt_context::TileKey key{ ctx->scene_version, ctx->camera_version, p->width, p->height, p->band_rows, p->band_first, p->band_step, p->tmin };
            const bool cached = ctx->tile_key_valid && memcmp(&key, &ctx->tile_key, sizeof key) == 0;
            TileArgs t{};
            t.geom = ctx->d_geom; t.aux = ctx->d_aux; t.n = ctx->n; t.cam = ctx->cam; t.tmin = p->tmin;
            t.ig.width = p->width; t.ig.height = p->height; t.ig.band_rows = p->band_rows; t.ig.band_first = p->band_first; t.ig.band_step = p->band_step;
            t.ig.owned_rows = owned; t.ig.tiles_x = tiles_x; t.ig.Q = (uint32_t)Q; t.ig.s_begin = 0; t.ig.s_count = 1; t.ig.tile_order = nullptr;
            t.tiles = n_tiles; t.cls = ctx->d_tile_cls; t.order = ctx->d_tile_order;
            if (!cached) {
                hipLaunchKernelGGL(k_tile_class, dim3((n_tiles * 8u + 255u) / 256u), dim3(256), 0, st, t);
                hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, st, t);
                HIP_TRY(ctx, hipGetLastError());
                ctx->tile_key = key; ctx->tile_key_valid = true;
            }
        }
        for (int bi = 0, s0 = 0; bi < batches; ++bi, s0 += s_per) {
            const int sc = (p->spp - s0 < s_per) ? p->spp - s0 : s_per;
            if (p->mode == RT_MODE_WAVEFRONT) {
                const uint32_t total = (uint32_t)(Q * sc);
                const uint32_t P = total < (8u << 20) ? (total + 255u) / 256u * 256u : (8u << 20);
                if (ctx->wf_P < P) {
                    HIP_TRY(ctx, hipStreamSynchronize(st));
                    (void)hipFree(ctx->wf_ray); (void)hipFree(ctx->wf_pixsmp); (void)hipFree(ctx->wf_depth); (void)hipFree(ctx->wf_hit_t); (void)hipFree(ctx->wf_hit_idx);
                    (void)hipFree(ctx->wf_stacks); (void)hipFree(ctx->wf_q[0]); (void)hipFree(ctx->wf_q[1]); (void)hipFree(ctx->wf_n);
                    ctx->wf_ray = nullptr; ctx->wf_pixsmp = nullptr; ctx->wf_depth = nullptr; ctx->wf_hit_t = nullptr; ctx->wf_hit_idx = nullptr;
                    ctx->wf_stacks = nullptr; ctx->wf_q[0] = ctx->wf_q[1] = nullptr; ctx->wf_n = nullptr; ctx->wf_P = 0;
                    HIP_TRY(ctx, hipMalloc(&ctx->wf_ray, (size_t)P * 6 * sizeof(float)));
                    HIP_TRY(ctx, hipMalloc(&ctx->wf_pixsmp, (size_t)P * 3 * sizeof(uint32_t)));
                    HIP_TRY(ctx, hipMalloc(&ctx->wf_depth, (size_t)P * sizeof(int)));
                    HIP_TRY(ctx, hipMalloc(&ctx->wf_hit_t, (size_t)P * sizeof(float)));
                    HIP_TRY(ctx, hipMalloc(&ctx->wf_hit_idx, (size_t)P * sizeof(int)));
                    HIP_TRY(ctx, hipMalloc(&ctx->wf_stacks, (size_t)P * RT_MAX_DEPTH * sizeof(uint16_t)));
                    HIP_TRY(ctx, hipMalloc(&ctx->wf_q[0], (size_t)P * sizeof(uint32_t)));
                    HIP_TRY(ctx, hipMalloc(&ctx->wf_q[1], (size_t)P * sizeof(uint32_t)));
                    HIP_TRY(ctx, hipMalloc(&ctx->wf_n, 2 * sizeof(uint32_t)));
                    ctx->wf_P = P;
                }
                WfArgs w;
                w.sw.frec = ctx->d_frec; w.sw.prec = ctx->d_prec; w.sw.plane_s0 = ctx->plane_s0; w.sw.plane_rmin = ctx->plane_rmin; w.sw.geom = ctx->d_geom; w.sw.n = ctx->n; w.sw.n_pad = ctx->n_pad;
                w.mat = ctx->d_mat; w.aux = ctx->d_aux; w.cam = ctx->cam;
                w.ig.width = p->width; w.ig.height = p->height; w.ig.band_rows = p->band_rows; w.ig.band_first = p->band_first; w.ig.band_step = p->band_step;
                w.ig.owned_rows = owned; w.ig.tiles_x = tiles_x; w.ig.Q = (uint32_t)Q; w.ig.s_begin = s0; w.ig.s_count = sc; w.ig.tile_order = nullptr;
                w.max_depth = p->max_depth; w.tmin = p->tmin; w.k0 = (uint32_t)p->seed; w.k1 = (uint32_t)(p->seed >> 32);
                w.total_items = total; w.P = ctx->wf_P;
                w.ray = ctx->wf_ray; w.pixsmp = ctx->wf_pixsmp; w.depth = ctx->wf_depth; w.hit_t = ctx->wf_hit_t; w.hit_idx = ctx->wf_hit_idx; w.stacks = ctx->wf_stacks;
                w.work_counter = ctx->d_work; w.ray_counter = ctx->d_rays; w.samples = ctx->d_samples;
                const bool wl = ctx->n <= 4096, wp = ctx->plane_filter;      // intersect keeps only the 16-byte exact records in LDS
                const size_t wshmem = (wl ? (size_t)ctx->n * sizeof(float4) : 0) + (size_t)(kTraceBlock / 64) * kSuperChunkWords * 64 * sizeof(uint32_t);
                int& wbpc = ctx->wf_blocks_per_cu[(wl ? 1 : 0) + (wp ? 2 : 0)];
                auto ikern = wl ? (wp ? k_wf_intersect<true, true> : k_wf_intersect<true, false>) : (wp ? k_wf_intersect<false, true> : k_wf_intersect<false, false>);
                if (wbpc == 0) { int nb = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ikern, kTraceBlock, wshmem) != hipSuccess || nb <= 0) nb = 1; wbpc = nb; }
                const unsigned igrid_cap = (unsigned)(ctx->num_cu * wbpc);
                // queue 0 is filled by k_wf_init; then intersect/shade alternate between the two queues
                w.q_in = ctx->wf_q[1]; w.q_out = ctx->wf_q[0]; w.n_in = ctx->wf_n + 1; w.n_out = ctx->wf_n + 0;
                const uint32_t first = P < total ? P : total;
                hipLaunchKernelGGL(k_wf_init, dim3((first + 255) / 256), dim3(256), 0, st, w);
                HIP_TRY(ctx, hipGetLastError());
                int cur = 0;
                uint32_t live_upper = first;          // the queue never grows: slots are reused, not added
                for (int iter = 0;; ++iter) {
                    w.q_in = ctx->wf_q[cur]; w.q_out = ctx->wf_q[cur ^ 1]; w.n_in = ctx->wf_n + cur; w.n_out = ctx->wf_n + (cur ^ 1);
                    HIP_TRY(ctx, hipMemsetAsync(ctx->wf_n + (cur ^ 1), 0, sizeof(uint32_t), st));
                    unsigned ib = (live_upper + kTraceBlock - 1) / kTraceBlock; if (ib > igrid_cap) ib = igrid_cap; if (ib == 0) ib = 1;
                    hipLaunchKernelGGL(ikern, dim3(ib), dim3(kTraceBlock), wshmem, st, w);
                    hipLaunchKernelGGL(k_wf_shade, dim3((live_upper + kWfShadeBlock - 1) / kWfShadeBlock), dim3(kWfShadeBlock), 0, st, w);
                    HIP_TRY(ctx, hipGetLastError());
                    cur ^= 1;
                    if ((iter & 7) == 7) {              // the host steers the loop: read the live count every 8 bounces
                        uint32_t live = 0;
                        HIP_TRY(ctx, hipMemcpyAsync(&live, ctx->wf_n + cur, sizeof live, hipMemcpyDeviceToHost, st));
                        HIP_TRY(ctx, hipStreamSynchronize(st));
                        if (live == 0) break;
                        live_upper = live;
                    }
                }
                ResolveArgs r;
                r.samples = ctx->d_samples; r.accum = ctx->d_accum; r.Q = (uint32_t)Q; r.tiles_x = tiles_x; r.s_count = sc;
                r.first = (bi == 0); r.last = (bi == batches - 1); r.width = p->width; r.owned_rows = owned; r.spp = p->spp;
                r.rgb_mean = d_rgb_mean; r.irgb = d_irgb;
                hipLaunchKernelGGL(k_resolve, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, st, r);
                HIP_TRY(ctx, hipGetLastError());
                continue;
            }
            TraceArgs a;
            a.prec = ctx->d_prec; a.plane_s0 = ctx->plane_s0; a.plane_rmin = ctx->plane_rmin; a.frec = ctx->d_frec; a.geom = ctx->d_geom; a.mat = ctx->d_mat; a.aux = ctx->d_aux; a.n = ctx->n; a.n_pad = ctx->n_pad; a.cam = ctx->cam;
            a.width = p->width; a.height = p->height; a.max_depth = p->max_depth; a.tmin = p->tmin;
            a.k0 = (uint32_t)p->seed; a.k1 = (uint32_t)(p->seed >> 32);
            a.band_rows = p->band_rows; a.band_first = p->band_first; a.band_step = p->band_step; a.owned_rows = owned;
            a.tiles_x = tiles_x; a.Q = (uint32_t)Q; a.s_begin = s0; a.s_count = sc; a.total_items = (uint32_t)(Q * sc);
            a.samples = ctx->d_samples; a.ray_counter = ctx->d_rays; a.work_counter = ctx->d_work;
            a.stacks = ctx->d_stacks; a.stack_stride = (uint32_t)ctx->stacks_lanes; a.staging = ctx->d_staging;
            a.tile_order = ordered ? ctx->d_tile_order : nullptr;
            a.bvh.nodes = ctx->d_bvh_nodes; a.bvh.leaf_index = ctx->d_bvh_leaf; a.bvh.big = ctx->d_bvh_big; a.bvh.n_big = ctx->bvh_n_big; a.bvh.geom = ctx->d_geom;
            // small jobs take smaller chunks so that every resident wave gets several (tail balance)
            const uint64_t resident_waves = (uint64_t)ctx->num_cu * bpc * (kTraceBlock / 64);
            uint64_t chunk = (uint64_t)a.total_items / (resident_waves * 16) / 64 * 64;
            const uint64_t floor_items = (32768ull / (uint64_t)ctx->n_pad + 63) / 64 * 64;     // tiny scenes: keep per-chunk work above the hand-out overhead
            if (chunk < floor_items) chunk = floor_items;
            a.chunk_items = (int)(chunk < 64 ? 64 : (chunk > (uint64_t)kChunkItems ? (uint64_t)kChunkItems : chunk));
            a.resident_waves = (uint32_t)resident_waves;
            { const uint64_t m = floor_items < 64 ? 64 : floor_items; a.min_chunk_items = (uint32_t)(m < (uint64_t)a.chunk_items ? m : (uint64_t)a.chunk_items); }
            uint64_t want_blocks = ((uint64_t)a.total_items + (uint64_t)a.chunk_items * (kTraceBlock / 64) - 1) / ((uint64_t)a.chunk_items * (kTraceBlock / 64));
            const uint64_t cap_blocks = (uint64_t)ctx->num_cu * bpc;
            const dim3 grid((unsigned)(want_blocks < cap_blocks ? want_blocks : cap_blocks)), block(kTraceBlock);
            // the first chunk of every launched wave is implied by its index; the counter hands out what follows
            HIP_TRY(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ctx->d_work), (int)(grid.x * (unsigned)(kTraceBlock / 64) * (unsigned)a.chunk_items), 1, st));
            if (bvh && lds) hipLaunchKernelGGL((k_trace<true, true, false, true>), grid, block, shmem, st, a);
            else if (bvh) hipLaunchKernelGGL((k_trace<false, true, false, true>), grid, block, shmem, st, a);
            else if (reclds) hipLaunchKernelGGL((k_trace<true, true, true>), grid, block, shmem, st, a);
            else if (lds && plane) hipLaunchKernelGGL((k_trace<true, true>), grid, block, shmem, st, a);
            else if (lds) hipLaunchKernelGGL((k_trace<true, false>), grid, block, shmem, st, a);
            else if (plane) hipLaunchKernelGGL((k_trace<false, true>), grid, block, shmem, st, a);
            else hipLaunchKernelGGL((k_trace<false, false>), grid, block, shmem, st, a);
            HIP_TRY(ctx, hipGetLastError());
            ResolveArgs r;
            r.samples = ctx->d_samples; r.accum = ctx->d_accum; r.Q = (uint32_t)Q; r.tiles_x = tiles_x; r.s_count = sc;
            r.first = (bi == 0); r.last = (bi == batches - 1); r.width = p->width; r.owned_rows = owned; r.spp = p->spp;
            r.rgb_mean = d_rgb_mean; r.irgb = d_irgb;
            hipLaunchKernelGGL(k_resolve, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, st, r);
            HIP_TRY(ctx, hipGetLastError());
        }
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev1, st));
    ctx->last_stream = st; ctx->pending = true; ctx->last_samples = (uint64_t)rt_owned_pixels(p) * (uint64_t)p->spp;
    return RT_OK;
}

int rt_render_stats(rt_context* ctx, uint64_t* rays, float* kernel_ms)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "rt_render_stats: null context");
    if (!ctx->pending) return fail(ctx, RT_ERR_STATE, "rt_render_stats: no render has been enqueued");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
    if (kernel_ms) HIP_TRY(ctx, hipEventElapsedTime(kernel_ms, ctx->ev0, ctx->ev1));
    if (rays) {
        unsigned long long r = 0;
        HIP_TRY(ctx, hipMemcpyAsync(&r, ctx->d_rays, sizeof r, hipMemcpyDeviceToHost, ctx->last_stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->last_stream));
        *rays = r;
    }
    return RT_OK;
}

int rt_render_counters(rt_context* ctx, rt_counters* out)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "rt_render_counters: null context");
    if (!out) return fail(ctx, RT_ERR_INVALID, "rt_render_counters: null output");
    if (!ctx->pending) return fail(ctx, RT_ERR_STATE, "rt_render_counters: no render has been enqueued");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    unsigned long long c[2] = { 0, 0 };
    HIP_TRY(ctx, hipMemcpyAsync(c, ctx->d_rays, sizeof c, hipMemcpyDeviceToHost, ctx->last_stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->last_stream));
    out->rays_traced = c[0];
    out->intersection_tests = c[0] - c[1];
    // every Color() call beyond the first of a sample was made by a successful Scatter below the depth cap
    out->material_evaluations = c[0] - ctx->last_samples;
    return RT_OK;
}

int rt_render(rt_context* ctx, const rt_params* p, float* h_rgb_mean, int32_t* h_irgb, uint64_t* rays, float* kernel_ms)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "rt_render: null context");
    if (!params_ok(p)) return fail(ctx, RT_ERR_INVALID, "rt_render: invalid parameters");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t npx = (size_t)rt_owned_pixels(p);
    float* d_rgb = nullptr; int32_t* d_i = nullptr;
    // The device side of the host outputs is kept in the context between calls (band-by-band callers make
    // many small calls; an allocation and a free per call cost more than the band's kernels).
    if (h_rgb_mean && npx) {
        const size_t need = npx * 3 * sizeof(float);
        if (ctx->out_rgb_bytes < need) {
            (void)hipFree(ctx->d_out_rgb); ctx->d_out_rgb = nullptr; ctx->out_rgb_bytes = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->d_out_rgb, need)); ctx->out_rgb_bytes = need;
        }
        d_rgb = ctx->d_out_rgb;
    }
    if (h_irgb && npx) {
        const size_t need = npx * 3 * sizeof(int32_t);
        if (ctx->out_irgb_bytes < need) {
            (void)hipFree(ctx->d_out_irgb); ctx->d_out_irgb = nullptr; ctx->out_irgb_bytes = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->d_out_irgb, need)); ctx->out_irgb_bytes = need;
        }
        d_i = ctx->d_out_irgb;
    }
    int rc = rt_render_device(ctx, p, d_rgb, d_i, nullptr);
    if (rc == RT_OK) rc = rt_render_stats(ctx, rays, kernel_ms);
    if (rc == RT_OK && d_rgb) { hipError_t e = hipMemcpy(h_rgb_mean, d_rgb, npx * 3 * sizeof(float), hipMemcpyDeviceToHost); if (e != hipSuccess) rc = fail(ctx, RT_ERR_HIP, "rt_render: copy back: %s", hipGetErrorString(e)); }
    if (rc == RT_OK && d_i) { hipError_t e = hipMemcpy(h_irgb, d_i, npx * 3 * sizeof(int32_t), hipMemcpyDeviceToHost); if (e != hipSuccess) rc = fail(ctx, RT_ERR_HIP, "rt_render: copy back: %s", hipGetErrorString(e)); }
    return rc;
}

}  // extern "C"
