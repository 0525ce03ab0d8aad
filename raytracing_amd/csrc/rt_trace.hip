// RT_MODE_MEGAKERNEL and RT_MODE_BVH: the persistent kernel k_trace, the work order of its tiles, and the ordered
// per-pixel sum k_resolve (shared with the wavefront mode).
#include "rt_sweep.hpp"

// ============================================================================ persistent megakernel
// Work item = one sample of one pixel.  Items are numbered tile by tile over 8x8 pixel tiles,
//   item = (tile * s_count + s_rel) * 64 + (row_in_tile * 8 + col_in_tile),
// so the 64 items a fresh wave picks up are one tile at one sample (coherent primary rays) and
// both this kernel's stores and the resolve kernel's loads of the sample buffer are coalesced.
// A wave draws items from a chunk it reserves with one atomic on a global counter; whenever a
// lane's path ends it is handed the next item (ballot + prefix count), so every trip of the main
// loop runs one closest-hit sweep with all lanes busy until the work runs out.
//
// The sweep (closest_hit) is split in three:
//   filter  (hot, branch-free, records as scalar operands): a conservative sign test per ray-sphere
//            pair -- by default the distance of the centre to the vertical plane through the ray
//            (3 FMAs), alternatively the whole discriminant expanded about the world origin (8 FMAs) --
//            with a per-sphere and per-ray slack that dominates every rounding error of both this
//            form and the reference's.  The sign bits of 32 spheres are shifted into one register.
//   pass A  (per lane, a dozen survivors per ray): the second of those two tests on the survivors of the first.
//   pass B  (per lane, about two per ray): the spheres still standing are evaluated with the reference's own
//            unfused arithmetic (sphere.cpp:15-52); the accepted root of each and the running minimum
//            follow hittablelist.cpp:13-31, so the chosen sphere and t are bit-identical.
// A wave left with only a few live paths (the end of a launch) switches to closest_hit_few.
struct TraceArgs {
    const float4* prec;      // plane-filter records (cx, cz, 2^88 R_i, 0), R_i = r^2 + slack_i, same padding as frec
    const uint32_t* force; int n_forced;   // per 32 spheres: the ones the plane filter does not judge (see closest_hit), and their number
    float plane_s0, plane_rmin;   // plane filter: constant part of the per-ray slack, smallest R_i
    const float4* frec;      // discriminant-filter records (cx, cy, cz, |c|^2 - r^2 - slack), padded to a multiple of 8
    const uint2* arec; float mfma_g;       // matrix-pipe form of the plane filter: A operands per chunk of 32 spheres, per-ray allowance (closest_hit)
    const float4* geom; const float4* mat; const Aux* aux; int n, n_pad;
    CameraD cam;
    int width, height, max_depth; float tmin;
    uint32_t k0, k1;
    int band_rows, band_first, band_step, owned_rows;
    int tiles_x; uint32_t Q;              // padded pixel count = tiles * 64
    int share_first, share_step;          // tile share of the call (ItemGeometry)
    int s_begin, s_count;                 // sample batch
    uint32_t total_items;                 // Q * s_count  (< 2^31)
    int chunk_items;                      // items a wave reserves per atomic: a multiple of 64, at most kChunkItems
    uint32_t resident_waves;              // waves of this launch that run at the same time
    uint32_t min_chunk_items;             // smallest chunk taken near the end of the work (a multiple of 64)
    float* samples;                       // [total_items][3]
    unsigned long long* ray_counter;
    unsigned int* work_counter;
    float* staging;                       // per wave: kChunkItems staged rays of kStagedFloats floats (see the hand-out in k_trace)
    const uint32_t* tile_order;           // work order of the tiles (see decode_item); may be null
    BvhScene bvh;                         // RT_MODE_BVH only
    uint16_t* stacks; uint32_t stack_stride;   // hit-index stacks in HBM: entry k of lane g at stacks[k * stack_stride + g]
    const uint32_t* tile_lists;           // per image tile: the spheres its camera rays can meet (k_tile_lists); null: every camera ray is swept
    float* queue;                         // kQueue 1: the camera ray of every item of the batch, row i = item i, 32 bytes (k_primary);
                                          // kQueue 2: per segment of chunk_items items the rows stage_chunk leaves, 48 bytes each, from row segment * chunk_items on (k_primary_lists)
    uint32_t* seg_count; uint32_t n_segments;      // kQueue 2: rows written per segment
};

// ============================================================================ work order of the tiles
// A persistent wave keeps all its lanes busy only while there are items left to hand out; once the counter
// runs dry every wave drains the paths it still holds, and a wave with one 50-bounce path left costs as many
// issue slots per bounce as a full one.  What is in flight at that moment is whatever was started LAST, so
// tiles are worked on in order of how long their paths can get: tiles that show a glass sphere first (a path can
// bounce inside one up to the depth cap), then those that show a metal sphere, tiles that show only diffuse spheres next, tiles where every probe ray leaves the scene (one ray per
// sample) last.  The order changes scheduling only: a sample's value and its slot in the sample buffer
// depend on the image tile, not on when it is traced.
struct TileArgs {
    const float4* geom; const Aux* aux; int n;
    CameraD cam; ItemGeometry ig; float tmin;
    uint32_t tiles; uint32_t* cls; uint32_t* order; uint32_t* rank;      // order[position] = image tile, rank[image tile] = position
};

// Eight probe rays per tile (corners and edge midpoints of the tile's pixel rectangle, through the lens
// centre), one lane each; class = the largest over the probes of {0 miss, 1 lambertian, 2 metal, 3 dielectric}.
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
        if (hit >= 0) cls = a.aux[hit].type == RT_LAMBERTIAN ? 1u : a.aux[hit].type == RT_METAL ? 2u : 3u;
    }
    for (int off = 1; off < 8; off <<= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)cls, off); cls = o > cls ? o : cls; }
    if (tile < a.tiles && sub == 0u) a.cls[tile] = cls;
}

// Stable counting sort of the tiles by class, highest class first (one block).
constexpr int kTileClasses = 4;
__global__ void __launch_bounds__(1024) k_tile_order(const TileArgs a)
{
    __shared__ uint32_t s_count[kTileClasses];
    __shared__ uint32_t s_wave[16][kTileClasses];
    __shared__ uint32_t s_run[kTileClasses];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < kTileClasses) s_count[threadIdx.x] = 0;
    __syncthreads();
    uint32_t mine[kTileClasses] = {};
    for (uint32_t i = threadIdx.x; i < a.tiles; i += 1024u) {
        const uint32_t c = a.cls[i];
#pragma unroll
        for (int k = 0; k < kTileClasses; ++k) mine[k] += c == (uint32_t)k ? 1u : 0u;
    }
#pragma unroll
    for (int c = 0; c < kTileClasses; ++c) if (mine[c]) atomicAdd(&s_count[c], mine[c]);
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t run = 0; for (int c = kTileClasses - 1; c >= 0; --c) { s_run[c] = run; run += s_count[c]; } }
    __syncthreads();
    for (uint32_t base = 0; base < a.tiles; base += 1024u) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t c = i < a.tiles ? a.cls[i] : (uint32_t)kTileClasses;
        uint32_t rank = 0;
        for (uint32_t k = 0; k < (uint32_t)kTileClasses; ++k) {
            const unsigned long long m = __ballot(c == k);
            if (c == k) rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (lane == 0) s_wave[wave][k] = (uint32_t)__popcll(m);
        }
        __syncthreads();
        if (c < (uint32_t)kTileClasses) {
            uint32_t before = s_run[c];
            for (int w = 0; w < wave; ++w) before += s_wave[w][c];
            a.order[before + rank] = i; a.rank[i] = before + rank;
        }
        __syncthreads();
        if (threadIdx.x < kTileClasses) { uint32_t tot = 0; for (int w = 0; w < 16; ++w) tot += s_wave[w][threadIdx.x]; s_run[threadIdx.x] += tot; }
        __syncthreads();
    }
}

// ============================================================================ the spheres a tile's camera rays can meet
// The camera rays of one 8x8 tile differ only by the jitter inside their pixels and by the point of the lens they start from: every
// one of them joins a point of the lens, o + off, to a point F(s, t) = llc + s H + t V of the tile's patch of the focal plane
// (Camera::GetRay, camera.cpp:46-53: origin o + off, direction F - o - off, off = lens_radius (dx u + dy v), dx^2 + dy^2 < 1).
// So the rays form an hourglass around the line from o through the middle Fc of the patch, and only a handful of spheres reaches
// into it.  k_tile_lists writes, per tile, the ascending list of those spheres; k_trace then finds the closest hit of a camera ray
// over its tile's list with the reference's own arithmetic (sphere.cpp:15-52, hittablelist.cpp:21-29) instead of a sweep over all
// spheres.  A sphere outside the list must be one that the reference's test rejects for every ray of the tile, so the list is built
// to err on the side of inclusion:
//   * with |off| <= l (l = lens_radius times the largest stretch of (dx, dy) -> dx u + dy v) and |F - Fc| <= h (h = the largest
//     distance of a corner of the patch from Fc; u = float(i + xi) / nx lies in [i / nx, (i + 1) / nx] up to one rounding, which
//     a margin on s and t covers), the point of a ray at parameter x, (1 - x)(o + off) + x F, is within rho(x) = |1 - x| l + |x| h
//     of the axis point A(x) = o + x (Fc - o).  A ray that meets the sphere (c, R) has |A(x) - c| <= R + rho(x) for some x (x > 0
//     unless roots behind the origin count, tmin < 0).  With p the distance of c from the axis line, x* the parameter of its foot
//     (moved to 0, with p = |c - o|, when it lies behind the origin), D = |Fc - o| and k = (l + h) / D, that requires
//     p sqrt(1 - k^2) <= R + rho(x*), because rho changes by at most (l + h) per unit of x while |A(x) - c| grows like D |x - x*|;
//   * R: the float discriminant of sphere.cpp:17-25 can be positive for a line that passes up to sqrt(r^2 + 24 u (Dc^2 + r^2))
//     from the centre, Dc >= |origin - centre|, u = 2^-24 (see closest_hit_bvh);
//   * the test runs in double precision; the rounding of the float ray (four additions per component of the direction, one of
//     the origin) moves its points by less than the absolute margin 1e-5 (1 + |c| + |o| + what the direction's terms add up to).
// A comparison that involves a NaN counts as "may be hit".  More than 63 candidates, k >= 1/2 or a camera that is not finite:
// no list for the tile (count 0xffffffff), its camera rays are swept like every other ray.
constexpr int kTileListWords = 64;          // word 0: count, words 1 .. 63: sphere indices in ascending order
constexpr uint32_t kNoTileList = 0xffffffffu;
struct TileListArgs { const float4* geom; int n; CameraD cam; ItemGeometry ig; float tmin; uint32_t tiles; uint32_t* lists; };

__global__ void __launch_bounds__(256) k_tile_lists(const TileListArgs a)
{
    const uint32_t tile = blockIdx.x * 4u + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (tile >= a.tiles) return;
    uint32_t* out = a.lists + (size_t)tile * kTileListWords;
    const int ty = (int)(tile / (uint32_t)a.ig.tiles_x), tx = (int)(tile - (uint32_t)ty * (uint32_t)a.ig.tiles_x);
    const int i0 = tx * 8, i1 = (i0 + 8 < a.ig.width) ? i0 + 8 : a.ig.width;
    const int r0 = ty * 8, r1 = (r0 + 7 < a.ig.owned_rows) ? r0 + 7 : a.ig.owned_rows - 1;
    const int y0 = owned_row_to_y2(a.ig.band_rows, a.ig.band_first, a.ig.band_step, r0);
    const int y1 = owned_row_to_y2(a.ig.band_rows, a.ig.band_first, a.ig.band_step, r1);      // y0 <= y1: the mapping is increasing
    // u = float(i + xi) / nx with xi in [0, 1) (main.cpp:126-127) lies in [i / nx, (i + 1) / nx] up to one rounding
    const double eps = 1e-6;
    const double s0 = (double)i0 / (double)a.ig.width - eps, s1 = (double)i1 / (double)a.ig.width + eps;
    const double t0 = (double)(a.ig.height - 1 - y1) / (double)a.ig.height - eps, t1 = (double)(a.ig.height - y0) / (double)a.ig.height + eps;
    const CameraD& c = a.cam;
    auto focal = [&](double s, double t, double& x, double& y, double& z) {
        x = (double)c.llc.x + s * (double)c.horiz.x + t * (double)c.vert.x;
        y = (double)c.llc.y + s * (double)c.horiz.y + t * (double)c.vert.y;
        z = (double)c.llc.z + s * (double)c.horiz.z + t * (double)c.vert.z;
    };
    const double ox = (double)c.origin.x, oy = (double)c.origin.y, oz = (double)c.origin.z;
    double fx, fy, fz;
    focal(0.5 * (s0 + s1), 0.5 * (t0 + t1), fx, fy, fz);
    double h = 0.0;
    for (int q = 0; q < 4; ++q) {
        double x, y, z;
        focal((q & 1) ? s1 : s0, (q & 2) ? t1 : t0, x, y, z);
        const double d = sqrt((x - fx) * (x - fx) + (y - fy) * (y - fy) + (z - fz) * (z - fz));
        h = !(d <= h) ? d : h;                  // a NaN ends up here
    }
    double wx = fx - ox, wy = fy - oy, wz = fz - oz;
    const double D = sqrt(wx * wx + wy * wy + wz * wz);
    wx /= D; wy /= D; wz /= D;
    const double uu = (double)dot(c.u, c.u), vv2 = (double)dot(c.v, c.v), uv = fabs((double)dot(c.u, c.v));
    const double l = fabs((double)c.lens_radius) * sqrt((uu > vv2 ? uu : vv2) + uv) * 1.0001;
    const double kappa = (l + h) / D;
    auto norm1 = [](V3 v) { return fabs((double)v.x) + fabs((double)v.y) + fabs((double)v.z); };
    const double u24 = 5.9604644775390625e-08;
    const double terms = norm1(c.llc) + norm1(c.horiz) + norm1(c.vert) + norm1(c.origin) + fabs((double)c.lens_radius) * (norm1(c.u) + norm1(c.v));
    const bool usable = (D > 0.0) && (kappa < 0.5) && (terms < 1e30);      // false for anything that is not finite
    if (!usable) { if (lane == 0) out[0] = kNoTileList; return; }
    const double shrink = sqrt(1.0 - kappa * kappa);
    const double o1 = norm1(c.origin);
    const bool both_ways = a.tmin < 0.0f;          // roots behind the origin are accepted then
    uint32_t count = 0;
    for (int base = 0; base < a.n; base += 64) {
        const int k = base + lane;
        bool cand = false;
        if (k < a.n) {
            const float4 g = a.geom[k];            // (centre, r * r)
            const double vx = (double)g.x - ox, vy = (double)g.y - oy, vz = (double)g.z - oz;
            const double vv = vx * vx + vy * vy + vz * vz;
            double along = vx * wx + vy * wy + vz * wz;
            double p = sqrt(fmax(0.0, vv - along * along));
            if (!both_ways && along < 0.0) { p = sqrt(vv); along = 0.0; }
            const double x = along / D;
            const double rho = fabs(1.0 - x) * l + fabs(x) * h;
            const double r2 = (double)g.w, far = sqrt(vv) + l;
            const double reach = sqrt(r2 + 24.0 * u24 * (far * far + r2)) * 1.00001 + 1e-5 * (1.0 + fabs((double)g.x) + fabs((double)g.y) + fabs((double)g.z) + o1 + terms * (1.0 + fabs(x)));
            cand = !(p * shrink > reach + rho);
        }
        const unsigned long long m = __ballot(cand);
        const uint32_t pos = count + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (cand && pos < (uint32_t)(kTileListWords - 1)) out[1 + pos] = (uint32_t)k;
        count += (uint32_t)__popcll(m);
    }
    if (lane == 0) out[0] = count <= (uint32_t)(kTileListWords - 1) ? count : kNoTileList;
}

// Tile classes from the lists (see k_tile_class for what they are for): the list names every sphere a camera ray of the tile can
// meet, so no glass sphere between the eight probe rays of k_tile_class is overlooked.  A tile without a list counts as glass.
__global__ void __launch_bounds__(256) k_tile_class_from_lists(const uint32_t* lists, const Aux* aux, uint32_t tiles, uint32_t* cls)
{
    const uint32_t tile = blockIdx.x * 256u + threadIdx.x;
    if (tile >= tiles) return;
    const uint32_t* list = lists + (size_t)tile * kTileListWords;
    const uint32_t count = list[0];
    uint32_t c = count == kNoTileList ? 3u : 0u;
    for (uint32_t q = 0; q < (count == kNoTileList ? 0u : count); ++q) {
        const int type = aux[list[1 + q]].type;
        const uint32_t k = type == RT_LAMBERTIAN ? 1u : type == RT_METAL ? 2u : 3u;
        c = k > c ? k : c;
    }
    cls[tile] = c;
}

// ============================================================================ camera rays of a batch
// When every camera ray is swept like any other ray (no per-tile lists), nothing about it depends on the scene: the camera rays of
// a whole sample batch are generated by this kernel, one lane per item at full occupancy, and written to a queue in HBM in the order
// of the items -- row i = item i: origin, direction, pixel key, sample number, 32 bytes -- and k_trace only reserves chunks of that
// queue.  Generating them inside the persistent kernel (per chunk, into a staging area of the wave) kept two hundred instructions of
// Philox and camera arithmetic, their registers and their scalar constants in a kernel that is held to 80 registers, and stopped a
// wave's live paths while it staged: config 3 takes 72.6 ms that way (k_trace with 14 spilled registers), 66.0 + 3.9 ms this way
// (k_trace without a spill).  The queue is written and read once (32 + 32 bytes per camera ray); the loop it feeds is
// main.cpp:118-133, the ray is Camera::GetRay (camera.cpp:46-53).
constexpr int kQueueFloats = 8;
constexpr uint32_t kNoSample = 0xffffffffu;        // sample word of a row that holds no ray (padding pixel of an edge tile)
// A thread takes every gridDim.x * 256-th item: with sixteen items per thread instead of one the camera's constants and the
// item arithmetic's scalar part are set up once per sixteen rays, and the row-wise part of the item arithmetic runs on the scalar
// unit (decode_item_row): 3.2 against 3.9 ms for the 240 M camera rays of config 3; the queue's 7.7 GB alone would take 1.1 ms,
// a plain fill runs at 6.9 TB/s here.  Small batches keep enough blocks to fill the chip.
//
// The lens sample is drawn by rejection (RandomUnitInDisk, camera.cpp:31-44; one try in 4.7 fails) and the wave retries in place
// until its unluckiest lane is through: 3.6 tries, 2.5 Philox blocks for all 64 lanes.  Collecting the failed items of several rows
// on a list in LDS and giving 64 of them their further tries together (1.3 blocks per ray) was built and measured: 5.9 ms.  A
// row stored with holes leaves the L2 as partly written lines before the holes are filled (2 MB are written per XCD between two
// rows of a wave), and those, like the 32-byte pieces that fill them later, are far slower to write than whole lines; with the
// rows stored whole and the pieces stored over them 4.5 ms, with the pieces left out (a lower bound, wrong rays) 2.8 ms.
constexpr uint32_t kPrimaryItemsPerThread = 16, kPrimaryMinBlocks = 4096;
__global__ void __launch_bounds__(256) k_primary(const TraceArgs a)
{
    ItemGeometry ig; ig.width = a.width; ig.height = a.height; ig.band_rows = a.band_rows; ig.band_first = a.band_first; ig.band_step = a.band_step;
    ig.owned_rows = a.owned_rows; ig.tiles_x = a.tiles_x; ig.Q = a.Q; ig.s_begin = a.s_begin; ig.s_count = a.s_count; ig.tile_order = a.tile_order; ig.share_first = a.share_first; ig.share_step = a.share_step;
#pragma unroll 1
    for (uint32_t it = blockIdx.x * 256u + threadIdx.x; it < a.total_items; it += gridDim.x * 256u) {
        uint32_t gp = 0u, gs = kNoSample; int i = 0, j = 0;
        RayD gr; gr.o = mk(0, 0, 0); gr.d = mk(0, 0, 0);
        // (a wave's 64 items are one row of the numbering: blocks of 256 threads, batches of whole rows)
        if (decode_item_row(ig, (uint32_t)__builtin_amdgcn_readfirstlane((int)(it >> 6)), it & 63u, gp, gs, i, j)) gr = camera_ray<true>(a.cam, gp, gs, i, j, (float)a.width, (float)a.height, a.k0, a.k1);
        else gs = kNoSample;
        v4f ra, rb;
        ra.x = gr.o.x; ra.y = gr.o.y; ra.z = gr.o.z; ra.w = gr.d.x;
        rb.x = gr.d.y; rb.y = gr.d.z; rb.z = __uint_as_float(gp); rb.w = __uint_as_float(gs);
        v4f* row = reinterpret_cast<v4f*>(a.queue + (size_t)it * kQueueFloats);
        row[0] = ra; row[1] = rb;
    }
}

// Six waves per SIMD (three 512-thread blocks per CU) is what the LDS budget allows and what hides the latency of the exact
// passes; the register allocation is held to it explicitly (80 VGPRs; the allocator's own choice drifts between 77 and 96
// with small source changes).  The hierarchy walk with the scene in LDS fits two blocks per CU only, so it keeps more registers.
#ifdef RT_SECTION_TIMERS
#define RT_TRACE_WAVES(kBvh, kLds)      /* the timers' accumulators need the registers */
#else
#ifndef RT_WAVES
#define RT_WAVES 6
#endif
#define RT_TRACE_WAVES(kBvh, kLds) __attribute__((amdgpu_waves_per_eu((kBvh) && (kLds) ? 4 : RT_WAVES, (kBvh) && (kLds) ? 4 : RT_WAVES)))
#endif
#ifndef RT_WIDE_PHILOX
#define RT_WIDE_PHILOX 1
#endif
constexpr bool kWidePhilox = RT_WIDE_PHILOX != 0;     // the staging phase of k_trace has the registers for 64-bit products
constexpr int kUnresolved = -2;             // staging row 9: the staged ray is the camera ray itself (no list for its tile), not the path's second ray
// The camera rays of the items [g, claim_end) (whole rows of 64: one tile at one sample each), all lanes busy, written as rows of
// kStagedFloats floats from `staged` on; returns how many rows were written.  With per-tile sphere lists this is also the first
// level of Color() for those rays (see below): finished paths store their sample and leave no row.  Runs inside k_trace for the
// chunk a wave has reserved, and as the body of k_primary_lists for one segment of a batch.
template <bool kLds>
__device__ __forceinline__ uint32_t stage_chunk(const TraceArgs& a, const ItemGeometry& ig, uint32_t g, uint32_t claim_end, int lane, float fnx, float fny,
                                                const float4* s_geom, const float4* s_mat, const Aux* s_aux, float* staged,
                                                uint32_t& rays, uint32_t& misses, SecTimer& T)
{
    uint32_t kept = 0;
    // the list of the tile the current row belongs to: lane q holds candidate q (index and exact record), fetched when the tile changes
    uint32_t list_tile = 0xffffffffu, cnt = kNoTileList; int cand = 0; float4 crec = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (uint32_t row = g; row < claim_end; row += 64u) {        // 64 items: one tile at one sample
        const uint32_t it = row + (uint32_t)lane;
        uint32_t gp = 0xffffffffu, gs = 0u; int i = 0, j = 0;
        const uint32_t gstore = it;                    // the sample's slot is the item number
        RayD gr; gr.o = mk(0, 0, 0); gr.d = mk(0, 0, 0);
        bool keep = decode_item_row(ig, (uint32_t)__builtin_amdgcn_readfirstlane((int)(row >> 6)), (uint32_t)lane, gp, gs, i, j);         // padding pixels of edge tiles are dropped here
        if (keep) gr = camera_ray<kWidePhilox>(a.cam, gp, gs, i, j, fnx, fny, a.k0, a.k1);
        T.mark(8);
        // The first level of Color() for the camera ray, here, where the 64 lanes hold the rays of one tile: the closest
        // hit over the tile's list (all lanes on the same few spheres, records as scalar loads, ascending sphere order
        // with strict comparisons: HittableList::Hit's answer), then the shading step.  What is staged is the SECOND
        // ray of the path with the index of the sphere it left; a path that ends here stores its sample and takes no lane.
        int first = kUnresolved;
        if (a.tile_lists) {
            const uint32_t ts = (uint32_t)__builtin_amdgcn_readfirstlane((int)(row >> 6));
            const uint32_t work_tile = ts / (uint32_t)a.s_count;
            if (work_tile != list_tile) {
                list_tile = work_tile;
                const uint32_t tile = image_tile(ig, work_tile);
                const uint32_t* list = a.tile_lists + (size_t)tile * kTileListWords;
                cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)list[0]);
                if (cnt != kNoTileList) {
                    cand = (uint32_t)lane < cnt ? (int)list[1 + lane] : 0;
                    crec = a.geom[cand];            // (centre, r * r)
                }
            }
            if (cnt != kNoTileList) {
                const float qa = dot(gr.d, gr.d);
                float pre_t = FLT_MAX; int pre_k = -1;
                for (uint32_t q = 0; q < cnt; ++q) {
                    const int k = __builtin_amdgcn_readlane(cand, (int)q);
                    const V3 cq = mk(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, crec.x), (int)q)),
                                     __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, crec.y), (int)q)),
                                     __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, crec.z), (int)q)));
                    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, crec.w), (int)q));
                    const V3 oc = gr.o - cq;
                    const float b = dot(oc, gr.d);
                    const float cc = dot(oc, oc) - r2;
                    const float desc = b * b - qa * cc;
                    if (desc > 0.0f) {
                        const float sq = sqrtf(desc);
                        float temp = (-b - sq) / qa;
                        if (!(temp < pre_t && temp > a.tmin)) temp = (-b + sq) / qa;
                        if (temp < pre_t && temp > a.tmin) { pre_t = temp; pre_k = k; }
                    }
                }
                T.mark(9);
                rays += (uint32_t)__popcll(__ballot(keep)); misses += (uint32_t)__popcll(__ballot(keep && pre_k < 0));
                if (keep) {
                    int d0 = 0; V3 sample;
                    SecTimer untimed;            // (diagnostic build: the staging phase is timed as a whole)
                    KeyedUniformsT<kWidePhilox> UW; UW.k0 = a.k0; UW.k1 = a.k1;
                    const bool done = shade_step<kLds, true>(gr, d0, pre_t, pre_k, gp, gs, a.max_depth, UW, s_geom, s_mat, s_aux, a.geom, a.mat, a.aux, StridedStack{ nullptr, 0 }, sample, untimed);
                    first = pre_k;
                    if (done) {
                        float* o = a.samples + (size_t)gstore * 3;
                        o[0] = sample.x; o[1] = sample.y; o[2] = sample.z;
                        keep = false;
                    }
                }
            }
        }
        T.mark(10);
        const unsigned long long km = __ballot(keep);
        const uint32_t slot = kept + __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
        kept += (uint32_t)__popcll(km);
        if (keep) {
            float* rec = staged + (size_t)slot * kStagedFloats;
            v4f ra, rb, rc;
            ra.x = gr.o.x; ra.y = gr.o.y; ra.z = gr.o.z; ra.w = gr.d.x;
            rb.x = gr.d.y; rb.y = gr.d.z; rb.z = __uint_as_float(gp); rb.w = __uint_as_float(gs);
            rc.x = __uint_as_float(gstore); rc.y = __int_as_float(first); rc.z = 0.0f; rc.w = 0.0f;
            *reinterpret_cast<v4f*>(rec) = ra; *reinterpret_cast<v4f*>(rec + 4) = rb; *reinterpret_cast<v4f*>(rec + 8) = rc;
        }
    }
    return kept;
}

// With per-tile sphere lists the first level of Color() is part of what happens before k_trace as well: every wave of this kernel
// takes whole segments of a batch (chunk_items consecutive items: a few samples of one tile), runs stage_chunk on them -- camera
// rays, closest hit over the tile's list, first shading step, 64 lanes on rays of one kind -- and leaves the second rays of the
// paths that go on as dense rows at the start of the segment's part of the queue, their number in seg_count.  k_trace reserves
// segments instead of chunks of items.  No atomics: segments are dealt to waves by index, counters are summed per wave.
__global__ void __launch_bounds__(256) k_primary_lists(const TraceArgs a)
{
    const int lane = threadIdx.x & 63;
    const uint32_t n_waves = gridDim.x * 4u;
    ItemGeometry ig; ig.width = a.width; ig.height = a.height; ig.band_rows = a.band_rows; ig.band_first = a.band_first; ig.band_step = a.band_step;
    ig.owned_rows = a.owned_rows; ig.tiles_x = a.tiles_x; ig.Q = a.Q; ig.s_begin = a.s_begin; ig.s_count = a.s_count; ig.tile_order = a.tile_order; ig.share_first = a.share_first; ig.share_step = a.share_step;
    uint32_t rays = 0, misses = 0;
    SecTimer untimed;
    for (uint32_t seg = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (threadIdx.x >> 6))); seg < a.n_segments; seg += n_waves) {
        const uint32_t g = seg * (uint32_t)a.chunk_items;
        const uint32_t end = (a.total_items - g < (uint32_t)a.chunk_items) ? a.total_items : g + (uint32_t)a.chunk_items;
        const uint32_t kept = stage_chunk<false>(a, ig, g, end, lane, (float)a.width, (float)a.height, nullptr, nullptr, nullptr,
                                                 a.queue + (size_t)g * kStagedFloats, rays, misses, untimed);
        if (lane == 0) a.seg_count[seg] = kept;
    }
    if (lane == 0 && rays) { atomicAdd(a.ray_counter, (unsigned long long)rays); atomicAdd(a.ray_counter + 1, (unsigned long long)misses); }
}

// A word of LDS that other waves of the block change, read afresh (a plain LDS load, not one through the generic address space)
__device__ __forceinline__ uint32_t lds_word_now(const uint32_t* p)
{
    typedef volatile __attribute__((address_space(3))) uint32_t LdsWord;
    return *(LdsWord*)p;
}
template <bool kLds, bool kPlane, bool kBvh = false, bool kMfma = false, int kQueue = 0, bool kTiled = false>
__global__ void __launch_bounds__(trace_block(kMfma)) RT_TRACE_WAVES(kBvh, kLds) k_trace(const TraceArgs a)
{
    static_assert(!kTiled || (kMfma && !kLds && !kBvh), "operand tiles: matrix-pipe filter, scene records not in LDS");
    constexpr int kTraceBlock = trace_block(kMfma);       // (shadows the 512 of the other kernels inside this one)
#ifndef RT_ITEM_IN_LDS
#define RT_ITEM_IN_LDS 1
#endif
    constexpr bool kItemInLds = kMfma && RT_ITEM_IN_LDS != 0;
    extern __shared__ float4 smem[];
    // LDS: [geom n][mat n][aux n (8 B)] when kLds; geom in the packed form of ExactLds (centre, discriminant-filter constant)
    // for the sweep, in the plain form (centre, r*r) for the hierarchy walk.  With the matrix-pipe filter its A operands follow
    // (512 bytes per chunk of 32 spheres).
    // The end of a launch is shared by the block in the matrix-pipe variants ("the last paths of a block" in the main loop).  Its
    // control words come first, at addresses known when the kernel is compiled (no register holds them): [0] = (waves of this block
    // that can get no more items) << 16 | paths those waves hold, [1..3] = number of entries of the three rotating path tables,
    // [4 + w] = paths wave w has announced, [16 + w] = which bounce of the shared end wave w is in (kTailIdle: not in it).
    // (With operand tiles -- kTiled, see closest_hit -- the waves of a block sweep in step from the first trip to the last; there
    // the words count the waves that are done for good, [0], and remember which have said so, [4 + w].)
    constexpr bool kTail = kMfma && !kBvh && !kTiled && kTailPaths > 0;
    uint32_t* tail_ctl = reinterpret_cast<uint32_t*>(smem);
    float4* scene_lds = smem + (kTail || kTiled ? kTailCtlBytes / 16 : 0);
    float4* s_geom = scene_lds;
    float4* s_mat = scene_lds + (kLds ? a.n : 0);
    Aux* s_aux = reinterpret_cast<Aux*>(scene_lds + (kLds ? 2 * a.n : 0));
    const int a_words = kLds && kMfma ? ((a.n + 31) >> 5) * 64 : 0;       // uint2 entries
    uint2* s_arec = reinterpret_cast<uint2*>(reinterpret_cast<char*>(scene_lds) + (kLds ? (size_t)a.n * kLdsBytesPerSphere : 0));
    if (kLds) {
        for (int i = threadIdx.x; i < a.n; i += kTraceBlock) { const float4 g = a.geom[i]; s_geom[i] = make_float4(g.x, g.y, g.z, kBvh ? g.w : a.frec[i].w); s_mat[i] = a.mat[i]; s_aux[i] = a.aux[i]; }
        for (int i = threadIdx.x; i < a_words; i += kTraceBlock) s_arec[i] = a.arec[i];
    }
    ExactLds exact_lds{ s_geom, s_aux, nullptr }; exact_lds.arec = s_arec;
    // (the wave's number as a scalar: everything derived from it -- the first chunk, and with it the whole bookkeeping of the
    // hand-out -- is then uniform for the compiler too, and lives in scalar registers behind scalar branches)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    // Per-lane filter masks of one super-chunk: word w of lane l at mask_words[w * 64] (lanes contiguous).
    char* lds_after_scene = reinterpret_cast<char*>(scene_lds) + (kLds ? (size_t)a.n * kLdsBytesPerSphere : 0) + (size_t)a_words * sizeof(uint2);
    // (In RT_MODE_BVH the same area holds each lane's stack of pending nodes instead.)
    constexpr int kWordsPerLane = kBvh ? kBvhStack : kSuperChunkWords;
    // (with the item slots below, a wave's area is kWordsPerLane + 1 words per lane)
    uint32_t* mask_words = reinterpret_cast<uint32_t*>(lds_after_scene) + (size_t)wave * (kWordsPerLane + (kMfma ? 1 : 0)) * 64 + lane;
    // One more LDS word per lane behind the mask words (matrix-pipe variant: 3 KB per block, what is left under two blocks per CU):
    // the sample-buffer slot of the lane's path, written when the lane takes an item and read once when its path ends.  In a
    // register it was one of a dozen values the allocator kept in scratch memory, and every scratch store of the main loop went
    // out to HBM (18 of 48 bytes written per ray).
    // It sits right behind the wave's mask words, so that one address register serves both.
    uint32_t* item_slot = mask_words + kWordsPerLane * 64;
    uint32_t* tail_table = reinterpret_cast<uint32_t*>(lds_after_scene);       // [2 tables][kTailFields][kTailPaths], in the words of the filter masks
    constexpr uint32_t kTailIdle = 0xffffffffu;
    static_assert(!kTail || 16 + kTraceBlock / 64 <= kTailCtlBytes / 4, "control words of the shared end");
    // Whose hit-index stack the lane's path uses, during the shared end: only the first few lanes of a wave hold a path then, and
    // lane l keeps the number in the item slot of lane l + 32 (an address offset, no register)
    static_assert(!kTail || (kTailPaths + kTraceBlock / 64 - 1) / (kTraceBlock / 64) <= 32, "a wave's share of the pooled paths must fit its first 32 lanes");
    static_assert(!kTail || kTailPaths <= kFewPaths * (kTraceBlock / 64), "a wave's share of the pooled paths must stay with the sparse sweep: the tables lie in the mask words");
    static_assert(!kTail || 2 * kTailFields * kTailPaths <= kSuperChunkWords * 64, "both path tables lie in the mask words of the block's first wave");
    uint32_t* stack_owner = item_slot + 32;
    if ((kTail || kTiled) && threadIdx.x < kTailCtlBytes / 4) tail_ctl[threadIdx.x] = threadIdx.x < 4 ? 0u : kTailIdle;
    // operand tiles: two buffers behind the waves' mask and item words; the first tile of the first sweep is requested here
    uint2* tile_buffers = reinterpret_cast<uint2*>(lds_after_scene + (size_t)(kTraceBlock / 64) * (kWordsPerLane + (kMfma ? 1 : 0)) * 64 * sizeof(uint32_t));
    uint32_t tiles_done = 0u;                  // wave-uniform: tiles swept so far (tile g lies in buffer g & 1)
    if constexpr (kTiled) {
        SweepScene sw0; sw0.arec = a.arec; sw0.n = a.n;
        stage_operand_tile(sw0, tile_buffers, 0, 1u);
        exact_lds.arec = tile_buffers;
    }
    __syncthreads();
    // The recursion's attenuation chain is replayed innermost-first when a path reaches the sky, so
    // each lane keeps the indices of the spheres it bounced off.  One 2-byte store per bounce and a
    // short read-back per finished path: kept in HBM so that LDS does not limit occupancy (packing the
    // first levels into registers was measured 1.5 % slower: the loads are hidden by the other waves).
    // The stack of a lane is a column of a.stacks: level k of this thread at block_stacks[k * stack_stride + threadIdx.x] (a base the
    // block shares and the thread's number, which is in a register anyway: no per-lane pointer is held across the main loop).
    uint16_t* const block_stacks = a.stacks + (size_t)blockIdx.x * kTraceBlock;

    // Philox products in the shading of the main loop: one 64-bit multiply per product (half the multiplier work) where the
    // register allocation has room for the pairs -- with the matrix-pipe filter it has (config 3: 76.4 -> 74.9 ms); in the other
    // variants the pairs cost the kernel spilled registers and more than they gain.
#ifndef RT_MAIN_WIDE_PHILOX
#define RT_MAIN_WIDE_PHILOX 1
#endif
    KeyedUniformsT<(kMfma && RT_MAIN_WIDE_PHILOX != 0)> U; U.k0 = a.k0; U.k1 = a.k1; U.slot(0, 0, 0); U.cur = U4{ 0, 0, 0, 0 };
    RayD ray; ray.o = mk(0, 0, 0); ray.d = mk(0, 0, 1);
    int depth = 0; uint32_t item = 0, pix = 0, smp = 0;
    bool active = false;
    uint32_t rays = 0, misses = 0;             // wave-uniform: queries and sky misses of the whole wave
    uint32_t pool_next = 0, pool_end = 0, pool_base = 0;      // wave-uniform: the staged rows not yet handed out are pool_next - pool_base .. pool_end - pool_base
    uint32_t claim_end = 0;                                   // wave-uniform: end of the items this wave reserved last
    float* staged = kQueue == 2 ? a.queue : a.staging + ((size_t)blockIdx.x * (kTraceBlock / 64) + wave) * (kStagedFloats * kChunkItems);       // (kQueue 2: rows are addressed from the start of the queue, pool_base stays 0)
    bool exhausted = false;                    // wave-uniform
    bool first_chunk = true;                   // wave-uniform
    const float fnx = (float)a.width, fny = (float)a.height;
    SecTimer T; T.start();
    ItemGeometry ig; ig.width = a.width; ig.height = a.height; ig.band_rows = a.band_rows; ig.band_first = a.band_first; ig.band_step = a.band_step;
    ig.owned_rows = a.owned_rows; ig.tiles_x = a.tiles_x; ig.Q = a.Q; ig.s_begin = a.s_begin; ig.s_count = a.s_count; ig.tile_order = a.tile_order; ig.share_first = a.share_first; ig.share_step = a.share_step;
    SweepScene sw; sw.arec = a.arec; sw.mfma_g = a.mfma_g; sw.frec = a.frec; sw.prec = a.prec; sw.force = a.force; sw.n_forced = a.n_forced; sw.plane_s0 = a.plane_s0; sw.plane_rmin = a.plane_rmin; sw.geom = a.geom; sw.n = a.n; sw.n_pad = a.n_pad;

#ifdef RT_WAVE_TIMELINE
    const unsigned long long t_enter = __builtin_amdgcn_s_memrealtime(); unsigned long long t_dry = 0, trips_dry = 0, sparse_trips = 0, sparse_ticks = 0, t_trip = 0, t_first_sweep = 0, t_first_done = 0, t_shared = 0, shared_bounces = 0; bool was_sparse = false;
#endif
    for (;;) {
#ifdef RT_WAVE_TIMELINE
        if (exhausted) { if (t_dry == 0) t_dry = __builtin_amdgcn_s_memrealtime(); ++trips_dry; }
        { const unsigned long long now = __builtin_amdgcn_s_memrealtime(); if (was_sparse) { ++sparse_trips; sparse_ticks += now - t_trip; } t_trip = now; was_sparse = false; }
#endif
        // ---------------------------------------------------------------- hand out new items
        unsigned long long need = __ballot(!active);
        while (need != 0ull && !(exhausted && pool_next == pool_end)) {
            if constexpr (kQueue == 2) {
                if (pool_next == pool_end) {
                    // one segment of the queue: its rows are dense from the segment's first row on
                    unsigned int sg = 0;
                    if (first_chunk) { first_chunk = false; sg = blockIdx.x * (uint32_t)(kTraceBlock / 64) + (uint32_t)wave; }
                    else { if (lane == 0) sg = atomicAdd(a.work_counter, 1u); sg = __builtin_amdgcn_readfirstlane(sg); }
                    if (sg >= a.n_segments) { exhausted = true; break; }
                    pool_next = sg * (uint32_t)a.chunk_items;
                    pool_end = pool_next + (uint32_t)__builtin_amdgcn_readfirstlane((int)a.seg_count[sg]);
                    if (pool_next == pool_end) continue;           // nothing of this segment goes on
                }
            } else
            if (pool_next == pool_end) {
                unsigned int g = 0;
                // Chunks shrink towards the end of the work (down to a floor that keeps the hand-out cheap next to
                // the sweep) so that the waves run out of items at nearly the same time: with full-size chunks to the
                // last, a wave can start a whole chunk
                // (some twenty trips) just before the counter runs dry while its neighbours go idle.  What is left
                // is estimated from this wave's previous reservation and one chunk per resident wave since.
                uint32_t take = (uint32_t)a.chunk_items;
                {
                    const uint32_t seen = claim_end + a.resident_waves * take;
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
                claim_end = (a.total_items - g < take) ? a.total_items : g + take;
                if constexpr (kQueue == 1) {
                    pool_end = claim_end;          // the rows of the queue are the items themselves
                } else {
                // Generate the camera rays of the whole chunk now, all lanes busy, into this wave's staging
                // area in HBM; handing an item to a lane later is then three loads.  (Generating on demand
                // ran this code on every trip of the main loop for the third of the lanes that had finished.)
                T.mark(0);
#ifdef RT_STAGE_PRIO
                set_wave_priority(RT_STAGE_PRIO);
#endif
                const uint32_t kept = stage_chunk<kLds>(a, ig, g, claim_end, lane, fnx, fny, s_geom, s_mat, s_aux, staged, rays, misses, T);
                pool_end = g + kept;        // what is handed out is the staged rows, not the items
                T.mark(11);
                // the rows are read back by other lanes of this wave: drain the stores first
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            const uint32_t avail = pool_end - pool_next;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
            const uint32_t want = (uint32_t)__popcll(need);
            if constexpr (kQueue == 1) {
                if (!active && rank < avail) {
                    const uint32_t row = pool_next + rank;
                    const v4f* rec = reinterpret_cast<const v4f*>(a.queue + (size_t)row * kQueueFloats);
                    const v4f ra = rec[0], rb = rec[1];
                    smp = __float_as_uint(rb.w);
                    if (smp != kNoSample) {
                        ray.o = mk(ra.x, ra.y, ra.z); ray.d = mk(ra.w, rb.x, rb.y);
                        pix = __float_as_uint(rb.z);
                        if (kItemInLds) *item_slot = row; else item = row;          // the sample's slot is the item number
                        depth = 0;
                        active = true;
                    }
                }
            } else
            if (!active && rank < avail) {
                const uint32_t slot = pool_next + rank - pool_base;
                v4f ra, rb; v2f rc;
                staged_read(staged + (size_t)slot * kStagedFloats, ra, rb, rc);
                ray.o = mk(ra.x, ra.y, ra.z); ray.d = mk(ra.w, rb.x, rb.y);
                pix = __float_as_uint(rb.z); smp = __float_as_uint(rb.w);
                if (kItemInLds) *item_slot = __float_as_uint(rc.x); else item = __float_as_uint(rc.x);               // where the sample is stored
                // a staged second ray comes with the sphere its path left: the first entry of this lane's hit stack
                const int first = __float_as_int(rc.y);
                depth = 0;
                if (first != kUnresolved) { block_stacks[threadIdx.x] = (uint16_t)first; depth = 1; }
                active = true;
            }
            pool_next += (want < avail ? want : avail);
            need = __ballot(!active);
            // (a row of the queue can be empty -- a padding pixel -- so a lane may still be without a path here; it asks again)
            if (want <= avail && !(kQueue == 1 && need != 0ull)) break;
        }
        T.mark(0);
        // ---------------------------------------------------------------- the last paths of a block
        // When the work counter has run dry a wave drains the paths it holds: one with 20 paths left makes ordinary trips for
        // them while its eleven neighbours, done or nearly done, cannot help, and the launch ends when the last such wave does.
        // Instead the block's waves pool what they hold and deal it out afresh before every bounce: path s of the table goes to
        // wave s mod 12, which sweeps its few paths with the spheres spread over the lanes (closest_hit_few), shades them and
        // appends the ones that go on to the next table; one barrier per bounce.  A path keeps the hit-index stack of the thread
        // it started on (the stack stays where it is, in HBM; the path carries that thread's number).  Nothing about a path's
        // arithmetic changes: same rays, same draws, same order of products.
        uint32_t tail_bounce = kTailIdle;          // wave-uniform: the bounce of the block's shared end this trip is (not in it)
        if constexpr (kTail) {
            // (Everything here sits behind "this wave can get no more items" -- a flag the hand-out keeps anyway -- and the state of
            // the shared end lives in LDS, so the trips of a long launch carry no register for it.)
            if (exhausted && pool_next == pool_end) {
                // the wave's number as a scalar, formed here (as a per-lane value it is hoisted out of the main loop and costs a register there)
                uint32_t* mine = tail_ctl + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
                tail_bounce = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_word_now(mine + 16));
                if (tail_bounce == kTailIdle) {
                    // This wave says once that it is out of items, keeps the number of paths it still holds up to date, and goes
                    // to the block's barrier when it holds none, or when every wave of the block is out of items and together they
                    // hold at most kTailPaths paths.  Both numbers live in one word (a wave's paths are added with its
                    // announcement, so whoever sees all waves announced sees all their paths); once all have announced the word
                    // only falls, so every wave that tests it later decides the same way, and every wave of the block arrives at
                    // the first barrier exactly once.  Behind it no wave is in a sweep any more: the mask words are free for the tables.
                    const uint32_t nlive = (uint32_t)__popcll(__ballot(active));
                    if (lane == 0) {
                        const uint32_t published = mine[4];
                        if (published == kTailIdle) atomicAdd(tail_ctl, 0x10000u + nlive);
                        else if (nlive != published) atomicSub(tail_ctl, published - nlive);
                        mine[4] = nlive;
                    }
                    bool go = nlive == 0u;
                    if (!go) {
                        const uint32_t word = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_word_now(tail_ctl));
                        go = (word >> 16) == (uint32_t)(kTraceBlock / 64) && (word & 0xffffu) <= (uint32_t)kTailPaths;
                    }
                    if (go) {
                        __syncthreads();
                        const unsigned long long lm = __ballot(active);
                        uint32_t base = 0u;
                        if (lm != 0ull) {
                            if (lane == 0) base = atomicAdd(&tail_ctl[1], (uint32_t)__popcll(lm));
                            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                        }
                        if (active) {
                            uint32_t* e = tail_table + base + __builtin_amdgcn_mbcnt_hi((uint32_t)(lm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lm, 0u));
                            e[0 * kTailPaths] = __float_as_uint(ray.o.x); e[1 * kTailPaths] = __float_as_uint(ray.o.y); e[2 * kTailPaths] = __float_as_uint(ray.o.z);
                            e[3 * kTailPaths] = __float_as_uint(ray.d.x); e[4 * kTailPaths] = __float_as_uint(ray.d.y); e[5 * kTailPaths] = __float_as_uint(ray.d.z);
                            e[6 * kTailPaths] = pix; e[7 * kTailPaths] = smp; e[8 * kTailPaths] = (uint32_t)depth | (threadIdx.x << 8);
                            e[9 * kTailPaths] = kItemInLds ? *item_slot : item;
                        }
                        active = false;
                        if (lane == 0) mine[16] = 0u;
                        __syncthreads();
                        tail_bounce = 0u;
#ifdef RT_WAVE_TIMELINE
                        t_shared = __builtin_amdgcn_s_memrealtime();
#endif
                    }
                }
                if (tail_bounce != kTailIdle) {
                    // bounce b reads table b % 2, whose entry count is word 1 + b % 3; all waves of the block read the same count
                    // after the same barrier, so they leave together
                    const uint32_t c_cur = tail_bounce % 3u;
                    const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_word_now(&tail_ctl[1u + c_cur]));
                    if (n == 0u) break;
#ifdef RT_WAVE_TIMELINE
                    ++shared_bounces;
#endif
                    if (threadIdx.x == 0) tail_ctl[1u + (c_cur + 2u) % 3u] = 0u;          // read two bounces ago, appended to in the next one
                    const uint32_t* src = tail_table + (tail_bounce & 1u) * (uint32_t)(kTailFields * kTailPaths);
                    const uint32_t slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) + (uint32_t)(kTraceBlock / 64) * (uint32_t)lane;
                    if (slot < n) {
                        ray.o = mk(__uint_as_float(src[0 * kTailPaths + slot]), __uint_as_float(src[1 * kTailPaths + slot]), __uint_as_float(src[2 * kTailPaths + slot]));
                        ray.d = mk(__uint_as_float(src[3 * kTailPaths + slot]), __uint_as_float(src[4 * kTailPaths + slot]), __uint_as_float(src[5 * kTailPaths + slot]));
                        pix = src[6 * kTailPaths + slot]; smp = src[7 * kTailPaths + slot];
                        const uint32_t dw = src[8 * kTailPaths + slot];
                        depth = (int)(dw & 0xffu);
                        *stack_owner = dw >> 8;
                        if (kItemInLds) *item_slot = src[9 * kTailPaths + slot]; else item = src[9 * kTailPaths + slot];
                        active = true;
                    }
                    if (lane == 0) mine[16] = tail_bounce + 1u;          // (this wave's own word: read at the top of its next trip)
                }
            }
        }
        if constexpr (kTiled) {
            // Every wave of the block makes every trip, with or without paths (the sweep's barriers need all twelve), until all of
            // them are done for good.  A wave says so once; the count is read behind the barrier that opens the sweep, and no wave
            // can add to it between that barrier and the next trip's (it would have to finish this sweep first, which needs every
            // wave at the barriers inside it), so all twelve read the same number and leave together.
            if (exhausted && pool_next == pool_end && !__any(active)) {
                if (lane == 0 && tail_ctl[4 + wave] == kTailIdle) { tail_ctl[4 + wave] = 0u; atomicAdd(tail_ctl, 1u); }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the first tile of this sweep, requested during the last super-chunk of the sweep before
            __syncthreads();
            ++tiles_done;
            if ((uint32_t)__builtin_amdgcn_readfirstlane((int)lds_word_now(tail_ctl)) == (uint32_t)(kTraceBlock / 64)) break;
        } else
        if (!__any(active)) {
            if (kTail && tail_bounce != kTailIdle) { __syncthreads(); continue; }
            if (exhausted && pool_next == pool_end) break;
            continue;
        }

        // A path of L bounces needs L trips of its wave, and with six waves taking turns on a SIMD a trip lasts
        // six times its issue time: a 50-bounce path started late keeps the whole launch waiting.  Waves that
        // hold a deep path are therefore given issue priority over their neighbours (same total work, the long
        // chains just finish sooner).
        // With the matrix-pipe filter every wave spends its filter loop at priority 0 (closest_hit) and everything else one level
        // up, deep paths further up still.
#ifndef RT_PRIO_BASE
#define RT_PRIO_BASE 1
#endif
        const int deepest = __any(active && depth >= RT_PRIO_3) ? 3 : __any(active && depth >= RT_PRIO_2) ? 2 : __any(active && depth >= RT_PRIO_1) ? 1 : 0;
        const int prio_other = kMfma ? (deepest + RT_PRIO_BASE > 3 ? 3 : deepest + RT_PRIO_BASE) : deepest;
        set_wave_priority(prio_other);
        float closest = FLT_MAX; int hit = -1;
        const unsigned long long live = __ballot(active);
#ifdef RT_WAVE_TIMELINE
        was_sparse = __popcll(live) <= kFewPaths && live != 0ull;
        if (t_first_sweep == 0) t_first_sweep = __builtin_amdgcn_s_memrealtime();
        else if (t_first_done == 0) t_first_done = __builtin_amdgcn_s_memrealtime();
#endif
        if (kBvh) { if (active) closest_hit_bvh<kLds>(ray, a.tmin, a.bvh, exact_lds, mask_words, closest, hit); }
        else if constexpr (kTiled) closest_hit<kLds, true, true, true, true>(ray, a.tmin, sw, exact_lds, mask_words, closest, hit, T, active, prio_other, &tiles_done);       // every wave, every trip
        else if (__popcll(live) <= kFewPaths) closest_hit_few<kLds>(ray, a.tmin, sw, exact_lds, live, lane, closest, hit);
        else if constexpr (kMfma) closest_hit<kLds, true, true, true>(ray, a.tmin, sw, exact_lds, mask_words, closest, hit, T, active, prio_other);     // the whole wave enters
        else if (active) closest_hit<kLds, kPlane, true>(ray, a.tmin, sw, exact_lds, mask_words, closest, hit, T);
        rays += (uint32_t)__popcll(live); misses += (uint32_t)__popcll(__ballot(active && hit < 0));
        if (active) {
            T.mark(4);
            // ------------------------------------------------------------ shade (main.cpp:13-39)
            V3 sample;
            ColumnStack path_stack{ block_stacks, threadIdx.x, a.stack_stride };
            if constexpr (kTail) {
                // (asked of the wave's own word again rather than carried across the sweep; it holds the next bounce by now)
                if (exhausted && pool_next == pool_end &&
                    lds_word_now(tail_ctl + 16 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) != kTailIdle)
                    path_stack.column = *stack_owner;
            }
            const bool done = shade_step<kLds>(ray, depth, closest, hit, pix, smp, a.max_depth, U, s_geom, s_mat, s_aux, a.geom, a.mat, a.aux, path_stack, sample, T);
            if (done) {
                float* o = a.samples + (size_t)(kItemInLds ? *item_slot : item) * 3;
                o[0] = sample.x; o[1] = sample.y; o[2] = sample.z;
                active = false;
            }
            T.mark(7);
        }
        if constexpr (kTail) {
            if (exhausted && pool_next == pool_end) {
                // (the bounce this trip was is one less than what the wave's word says now)
                const uint32_t next_bounce = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_word_now(tail_ctl + 16 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))));
                if (next_bounce != kTailIdle) {
                    // the paths that go on: into the other table, behind what the other waves have put there
                    const unsigned long long gm = __ballot(active);
                    if (gm != 0ull) {
                        uint32_t base = 0u;
                        if (lane == 0) base = atomicAdd(&tail_ctl[1u + next_bounce % 3u], (uint32_t)__popcll(gm));
                        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                        if (active) {
                            uint32_t* e = tail_table + (next_bounce & 1u) * (uint32_t)(kTailFields * kTailPaths) + base + __builtin_amdgcn_mbcnt_hi((uint32_t)(gm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)gm, 0u));
                            e[0 * kTailPaths] = __float_as_uint(ray.o.x); e[1 * kTailPaths] = __float_as_uint(ray.o.y); e[2 * kTailPaths] = __float_as_uint(ray.o.z);
                            e[3 * kTailPaths] = __float_as_uint(ray.d.x); e[4 * kTailPaths] = __float_as_uint(ray.d.y); e[5 * kTailPaths] = __float_as_uint(ray.d.z);
                            e[6 * kTailPaths] = pix; e[7 * kTailPaths] = smp;
                            e[8 * kTailPaths] = (uint32_t)depth | (*stack_owner << 8);
                            e[9 * kTailPaths] = kItemInLds ? *item_slot : item;
                        }
                        active = false;
                    }
                    __syncthreads();
                }
            }
        }
    }
    T.flush(lane);
#ifdef RT_WAVE_TIMELINE
    {
        const uint32_t w = blockIdx.x * (kTraceBlock / 64) + wave;
        if (lane == 0 && w < 8192u) { g_wave_times[0][w] = t_enter; const unsigned long long t_exit = __builtin_amdgcn_s_memrealtime(); g_wave_times[1][w] = t_dry ? t_dry : t_exit; g_wave_times[2][w] = t_exit; g_wave_times[3][w] = trips_dry; g_wave_times[4][w] = sparse_trips; g_wave_times[5][w] = sparse_ticks; g_wave_times[6][w] = t_first_sweep; g_wave_times[7][w] = t_first_done; g_wave_times[8][w] = t_shared; g_wave_times[9][w] = shared_bounces; }
    }
#endif
    // one atomic per wave
    if (lane == 0 && rays) { atomicAdd(a.ray_counter, (unsigned long long)rays); atomicAdd(a.ray_counter + 1, (unsigned long long)misses); }
}

// Ordered per-pixel sum of a sample batch: col += sample for s ascending, exactly the
// accumulation order of main.cpp:124-134; the last batch applies `col /= ns` and quantises.
struct ResolveArgs {
    const float* samples; float* accum;     // accum: [Q][3] running sums
    const uint32_t* tile_rank;              // position of every image tile in the work order (the order of the sample buffer); null: image order
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
    const float* sp = a.samples + 3 * ((size_t)(a.tile_rank ? a.tile_rank[tile] : tile) * a.s_count * 64 + l);       // item = (work tile * s_count + s) * 64 + l
    for (int s = 0; s < a.s_count; ++s, sp += 3 * 64) col = col + mk(sp[0], sp[1], sp[2]);
    if (!a.last) { a.accum[3 * (size_t)q] = col.x; a.accum[3 * (size_t)q + 1] = col.y; a.accum[3 * (size_t)q + 2] = col.z; return; }
    const float inv = 1.0f / (float)a.spp;          // Vec3::operator/=(float), vec3.cpp:109-118
    const float mr = col.x * inv, mg = col.y * inv, mb = col.z * inv;
    const size_t o = ((size_t)r * a.width + i) * 3;
    if (a.rgb_mean) { a.rgb_mean[o] = mr; a.rgb_mean[o + 1] = mg; a.rgb_mean[o + 2] = mb; }
    if (a.irgb) { a.irgb[o] = quantise(mr); a.irgb[o + 1] = quantise(mg); a.irgb[o + 2] = quantise(mb); }
}
// The same ordered sum for a tile share: output tile-major, [work tile][pixel of the 8x8 tile][3], padding pixels of edge tiles
// zero, and the image tile each work tile is (so that whoever assembles the image needs nothing but these two arrays).
struct ResolveTilesArgs {
    const float* samples; float* accum; uint32_t Q; int tiles_x; int s_count; int first, last;
    int width, height, spp; int share_first, share_step; const uint32_t* tile_order;
    float* rgb_mean; int32_t* irgb; uint32_t* tile_ids;
    unsigned long long* pixel_counter;      // += the share's pixels inside the image (last batch only)
};
__global__ void __launch_bounds__(256) k_resolve_tiles(const ResolveTilesArgs a)
{
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    if (q >= a.Q) return;
    const uint32_t wt = q >> 6, l = q & 63u;
    const uint32_t pos = (uint32_t)a.share_first + wt * (uint32_t)a.share_step;
    const uint32_t tile = a.tile_order ? a.tile_order[pos] : pos;
    const int ty = (int)(tile / (uint32_t)a.tiles_x), tx = (int)(tile - (uint32_t)ty * (uint32_t)a.tiles_x);
    const int i = tx * 8 + (int)(l & 7u), y = ty * 8 + (int)(l >> 3);
    const bool inside = i < a.width && y < a.height;
    V3 col = a.first || !inside ? mk(0.0f, 0.0f, 0.0f) : mk(a.accum[3 * (size_t)q], a.accum[3 * (size_t)q + 1], a.accum[3 * (size_t)q + 2]);
    if (inside) {
        const float* sp = a.samples + 3 * ((size_t)wt * a.s_count * 64 + l);
        for (int s = 0; s < a.s_count; ++s, sp += 3 * 64) col = col + mk(sp[0], sp[1], sp[2]);
    }
    if (!a.last) { a.accum[3 * (size_t)q] = col.x; a.accum[3 * (size_t)q + 1] = col.y; a.accum[3 * (size_t)q + 2] = col.z; return; }
    const float inv = 1.0f / (float)a.spp;
    const float mr = col.x * inv, mg = col.y * inv, mb = col.z * inv;
    const size_t o = (size_t)q * 3;
    if (a.rgb_mean) { a.rgb_mean[o] = inside ? mr : 0.0f; a.rgb_mean[o + 1] = inside ? mg : 0.0f; a.rgb_mean[o + 2] = inside ? mb : 0.0f; }
    if (a.irgb) { a.irgb[o] = inside ? quantise(mr) : 0; a.irgb[o + 1] = inside ? quantise(mg) : 0; a.irgb[o + 2] = inside ? quantise(mb) : 0; }
    if (a.tile_ids && l == 0u) a.tile_ids[wt] = tile;
    const unsigned long long m = __ballot(inside);
    if ((threadIdx.x & 63u) == 0u && m != 0ull) atomicAdd(a.pixel_counter, (unsigned long long)__popcll(m));
}
#ifdef RT_SECTION_TIMERS
extern "C" int rt_debug_section_ticks(unsigned long long* out8, int reset)   /* kSections (12) values */
{
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_section_ticks), kSections * sizeof(unsigned long long)) != hipSuccess) return -3;
    if (reset) { unsigned long long z[kSections] = { 0 }; if (hipMemcpyToSymbol(HIP_SYMBOL(g_section_ticks), z, sizeof z) != hipSuccess) return -3; }
    return 0;
}
#endif
#ifdef RT_WAVE_TIMELINE
extern "C" int rt_debug_wave_times(unsigned long long* out_6x8192)   /* 10 rows of 8192 */
{
    return hipMemcpyFromSymbol(out_6x8192, HIP_SYMBOL(g_wave_times), 10 * 8192 * sizeof(unsigned long long)) == hipSuccess ? 0 : -3;
}
#endif

namespace {
using TraceKernel = void (*)(const TraceArgs);
TraceKernel trace_kernel(const TracePlan& t)
{
    if (t.tiled) return t.queue == 2 ? k_trace<false, true, false, true, 2, true> : k_trace<false, true, false, true, 1, true>;
    if (t.queue == 1) {
        if (t.mfma) return t.lds ? k_trace<true, true, false, true, 1> : k_trace<false, true, false, true, 1>;
        if (t.bvh) return t.lds ? k_trace<true, true, true, false, 1> : k_trace<false, true, true, false, 1>;
        if (t.lds) return t.plane ? k_trace<true, true, false, false, 1> : k_trace<true, false, false, false, 1>;
        return t.plane ? k_trace<false, true, false, false, 1> : k_trace<false, false, false, false, 1>;
    }
    if (t.queue == 2) {
        if (t.mfma) return t.lds ? k_trace<true, true, false, true, 2> : k_trace<false, true, false, true, 2>;
        if (t.bvh) return t.lds ? k_trace<true, true, true, false, 2> : k_trace<false, true, true, false, 2>;
        if (t.lds) return t.plane ? k_trace<true, true, false, false, 2> : k_trace<true, false, false, false, 2>;
        return t.plane ? k_trace<false, true, false, false, 2> : k_trace<false, false, false, false, 2>;
    }
    if (t.mfma) return t.lds ? k_trace<true, true, false, true> : k_trace<false, true, false, true>;
    if (t.bvh) return t.lds ? k_trace<true, true, true> : k_trace<false, true, true>;
    if (t.lds) return t.plane ? k_trace<true, true> : k_trace<true, false>;
    return t.plane ? k_trace<false, true> : k_trace<false, false>;
}
}  // namespace

// The candidate lists of every tile of the rows `p` selects, into ctx->d_tile_lists.  Rebuilt with every call (it is part of the
// render, and cheap next to it): nothing to invalidate when scene or camera change.
static int build_tile_lists(rt_context* ctx, const rt_params* p, int owned, hipStream_t st)
{
    const int tiles_x = (p->width + 7) / 8;
    const uint32_t n_tiles = (uint32_t)tiles_x * (uint32_t)((owned + 7) / 8);
    if (ctx->tile_list_cap < n_tiles) {
        HIP_TRY(ctx, hipStreamSynchronize(st));
        (void)hipFree(ctx->d_tile_lists); ctx->d_tile_lists = nullptr; ctx->tile_list_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_tile_lists, (size_t)n_tiles * kTileListWords * sizeof(uint32_t)));
        ctx->tile_list_cap = n_tiles;
    }
    TileListArgs la{};
    la.geom = ctx->d_geom; la.n = ctx->n; la.cam = ctx->cam; la.tmin = p->tmin;
    la.ig.width = p->width; la.ig.height = p->height; la.ig.band_rows = p->band_rows; la.ig.band_first = p->band_first; la.ig.band_step = p->band_step;
    la.ig.owned_rows = owned; la.ig.tiles_x = tiles_x; la.ig.Q = n_tiles * 64u; la.ig.s_begin = 0; la.ig.s_count = 1; la.ig.tile_order = nullptr;
    la.tiles = n_tiles; la.lists = ctx->d_tile_lists;
    hipLaunchKernelGGL(k_tile_lists, dim3((n_tiles + 3u) / 4u), dim3(256), 0, st, la);
    HIP_TRY(ctx, hipGetLastError());
    return RT_OK;
}

extern "C" int rt_test_tile_lists(rt_context* ctx, const rt_params* p, uint32_t* lists, int64_t capacity_words, int64_t* tiles)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "rt_test_tile_lists: null context");
    if (!params_ok(p) || !tiles) return fail(ctx, RT_ERR_INVALID, "rt_test_tile_lists: invalid parameters");
    if (!ctx->n || !ctx->have_camera) return fail(ctx, RT_ERR_STATE, "rt_test_tile_lists: no scene or no camera");
    const int64_t owned = rt_owned_rows(p);
    *tiles = (int64_t)((p->width + 7) / 8) * ((owned + 7) / 8);
    if (*tiles == 0) return RT_OK;
    if (*tiles > (1ll << 24)) return fail(ctx, RT_ERR_TOO_LARGE, "rt_test_tile_lists: %lld tiles", (long long)*tiles);
    if (!lists || capacity_words < *tiles * kTileListWords) return fail(ctx, RT_ERR_TOO_LARGE, "rt_test_tile_lists: %lld words needed", (long long)(*tiles * kTileListWords));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // a render still in flight on another stream reads the lists this call is about to rebuild (and may free)
    if (ctx->pending) HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
    const int rc = build_tile_lists(ctx, p, (int)owned, nullptr);
    if (rc != RT_OK) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(nullptr));
    HIP_TRY(ctx, hipMemcpy(lists, ctx->d_tile_lists, (size_t)*tiles * kTileListWords * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return RT_OK;
}

// Settles what every batch of the call shares: which instantiation of k_trace runs and how many of its blocks fit a CU,
// the per-lane hit stacks and per-wave staging rows for that many resident waves, and the work order of the tiles.
int trace_prepare(rt_context* ctx, const RenderFrame& f, TracePlan* plan)
{
    const rt_params* p = f.p;
    hipStream_t st = f.st;
    TracePlan& t = *plan;
    t.lds = ctx->n <= (ctx->max_lds_spheres > 0 ? ctx->max_lds_spheres : kMaxLdsSpheres);
    t.plane = ctx->plane_filter;
    t.bvh = p->mode == RT_MODE_BVH;
    // (a matrix-pipe chunk costs the same for 1 sphere as for 32, about what the vector form spends on 20: tiny scenes stay on the
    // vector ALU -- the four spheres of the three-sphere test scene: 1.40 against 1.58 ms at 400x225x100)
    t.mfma = !t.bvh && t.plane && ctx->mfma_filter && ctx->n >= 20;
    t.block = trace_block(t.mfma);
    t.queue = !ctx->primary_kernel ? 0 : ctx->primary_lists ? 2 : 1;
    // With the matrix-pipe filter a block keeps 40 + 16 bytes per sphere and 48 KB of mask words in LDS: two blocks per CU (six waves
    // per SIMD) up to 585 spheres, one above.  One block per CU loses more than LDS-resident records gain (600 spheres: 4.98 ms with
    // the records in LDS, 4.17 ms with them read from L2 at full occupancy; 1 536 spheres: 9.89 / 8.18 ms; tools/lds_threshold.py), so
    // the records stay in LDS only while two blocks fit.
    if (t.mfma && t.lds && ctx->max_lds_spheres == 0 &&
        (size_t)ctx->n * kLdsBytesPerSphere + (size_t)((ctx->n + 31) / 32) * 512 + (size_t)(t.block / 64) * kSuperChunkWords * 64 * sizeof(uint32_t) + (size_t)t.block * sizeof(uint32_t) + (kTailPaths > 0 ? kTailCtlBytes : 0) > 80u * 1024u) t.lds = false;
    // Scenes whose records do not fit LDS: every wave streams the matrix-pipe filter's operands from L2 for itself (default), or
    // -- RT_AMD_LARGE_SCENES=tiles when the context is created -- a block stages them in LDS as tiles of 512 spheres for its twelve
    // waves, which then sweep in step (closest_hit, kTiled).  Same images; 20 000 spheres at 1920x1080x32: 197 ms streamed, 225 ms
    // tiled (the barrier per tile costs the waves more than the shared operands save: L2 serves the stream at a 99 % hit rate)
    t.tiled = t.mfma && !t.lds && t.queue != 0 && ctx->operand_tiles;
    t.shmem = (t.lds ? (size_t)ctx->n * kLdsBytesPerSphere + (t.mfma ? (size_t)((ctx->n + 31) / 32) * 512 : 0) : 0) + (size_t)(t.block / 64) * (t.bvh ? kBvhStack : kSuperChunkWords) * 64 * sizeof(uint32_t) + (t.mfma ? (size_t)t.block * sizeof(uint32_t) + (kTailPaths > 0 || t.tiled ? kTailCtlBytes : 0) : 0) + (t.tiled ? 2u * kTileEntries * sizeof(uint2) : 0);
    int& bpc = ctx->trace_blocks_per_cu[t.queue * 8 + (t.tiled ? 3 : t.mfma ? (t.lds ? 7 : 6) : t.bvh ? (t.lds ? 5 : 4) : (t.lds ? 1 : 0) + (t.plane ? 2 : 0))];
    if (bpc == 0) {
        int nb = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, trace_kernel(t), t.block, t.shmem);
        if (e != hipSuccess) return fail(ctx, RT_ERR_HIP, "occupancy query for the render kernel (%zu bytes of LDS per block) failed: %s", t.shmem, hipGetErrorString(e));
        if (nb <= 0) return fail(ctx, RT_ERR_TOO_LARGE, "the render kernel does not fit a compute unit with %zu bytes of LDS per block (%d spheres)", t.shmem, ctx->n);
        bpc = nb;
    }
    t.blocks_per_cu = bpc;
    const size_t lanes = (size_t)ctx->num_cu * bpc * t.block;
    if (ctx->stacks_lanes < lanes) {
        HIP_TRY(ctx, hipStreamSynchronize(st));
        (void)hipFree(ctx->d_stacks); ctx->d_stacks = nullptr; ctx->stacks_lanes = 0;
        (void)hipFree(ctx->d_staging); ctx->d_staging = nullptr;
        HIP_TRY(ctx, hipMalloc(&ctx->d_stacks, lanes * RT_MAX_DEPTH * sizeof(uint16_t)));
        HIP_TRY(ctx, hipMalloc(&ctx->d_staging, (lanes / 64) * kStagedFloats * kChunkItems * sizeof(float)));
        ctx->stacks_lanes = lanes;
    }
    if (t.queue) {
        const size_t need = (size_t)f.Q * (size_t)f.s_per * (t.queue == 2 ? kStagedFloats : kQueueFloats) * sizeof(float);
        if (ctx->queue_bytes < need) {
            HIP_TRY(ctx, hipStreamSynchronize(st));
            (void)hipFree(ctx->d_queue); ctx->d_queue = nullptr; ctx->queue_bytes = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->d_queue, need)); ctx->queue_bytes = need;
        }
    }
    const uint32_t n_tiles = f.image_tiles;          // classes, order and lists cover the whole image also when the call renders a share of its tiles
    if (ctx->primary_lists) { const int rc = build_tile_lists(ctx, p, f.owned, st); if (rc != RT_OK) return rc; }
    t.ordered = ctx->tile_ordering;
    if (!t.ordered) return RT_OK;
    if (ctx->tile_cap < n_tiles) {
        HIP_TRY(ctx, hipStreamSynchronize(st));
        (void)hipFree(ctx->d_tile_cls); (void)hipFree(ctx->d_tile_order); (void)hipFree(ctx->d_tile_rank); ctx->d_tile_cls = ctx->d_tile_order = ctx->d_tile_rank = nullptr; ctx->tile_cap = 0; ctx->tile_key_valid = false;
        HIP_TRY(ctx, hipMalloc(&ctx->d_tile_cls, (size_t)n_tiles * sizeof(uint32_t)));
        HIP_TRY(ctx, hipMalloc(&ctx->d_tile_order, (size_t)n_tiles * sizeof(uint32_t)));
        HIP_TRY(ctx, hipMalloc(&ctx->d_tile_rank, (size_t)n_tiles * sizeof(uint32_t)));
        ctx->tile_cap = n_tiles;
    }
    const rt_context::TileKey key{ ctx->scene_version, ctx->camera_version, p->width, p->height, p->band_rows, p->band_first, p->band_step, p->tmin };
    if (ctx->tile_key_valid && memcmp(&key, &ctx->tile_key, sizeof key) == 0) return RT_OK;
    TileArgs ta{};
    ta.geom = ctx->d_geom; ta.aux = ctx->d_aux; ta.n = ctx->n; ta.cam = ctx->cam; ta.tmin = p->tmin;
    ta.ig.width = p->width; ta.ig.height = p->height; ta.ig.band_rows = p->band_rows; ta.ig.band_first = p->band_first; ta.ig.band_step = p->band_step;
    ta.ig.owned_rows = f.owned; ta.ig.tiles_x = f.tiles_x; ta.ig.Q = n_tiles * 64u; ta.ig.s_begin = 0; ta.ig.s_count = 1; ta.ig.tile_order = nullptr;
    ta.tiles = n_tiles; ta.cls = ctx->d_tile_cls; ta.order = ctx->d_tile_order; ta.rank = ctx->d_tile_rank;
    if (ctx->primary_lists) hipLaunchKernelGGL(k_tile_class_from_lists, dim3((n_tiles + 255u) / 256u), dim3(256), 0, st, ctx->d_tile_lists, ctx->d_aux, n_tiles, ctx->d_tile_cls);
    else hipLaunchKernelGGL(k_tile_class, dim3((n_tiles * 8u + 255u) / 256u), dim3(256), 0, st, ta);
    hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, st, ta);
    HIP_TRY(ctx, hipGetLastError());
    ctx->tile_key = key; ctx->tile_key_valid = true;
    return RT_OK;
}

int trace_batch(rt_context* ctx, const RenderFrame& f, const TracePlan& t, const SampleBatch& b)
{
    const rt_params* p = f.p;
    TraceArgs a;
    a.arec = ctx->d_arec; a.mfma_g = ctx->mfma_g; a.prec = ctx->d_prec; a.force = ctx->d_force; a.n_forced = ctx->n_forced; a.plane_s0 = ctx->plane_s0; a.plane_rmin = ctx->plane_rmin; a.frec = ctx->d_frec; a.geom = ctx->d_geom; a.mat = ctx->d_mat; a.aux = ctx->d_aux; a.n = ctx->n; a.n_pad = ctx->n_pad; a.cam = ctx->cam;
    a.width = p->width; a.height = p->height; a.max_depth = p->max_depth; a.tmin = p->tmin;
    a.k0 = (uint32_t)p->seed; a.k1 = (uint32_t)(p->seed >> 32);
    a.band_rows = p->band_rows; a.band_first = p->band_first; a.band_step = p->band_step; a.owned_rows = f.owned;
    a.tiles_x = f.tiles_x; a.Q = (uint32_t)f.Q; a.share_first = f.share_first; a.share_step = f.share_step; a.s_begin = b.s0; a.s_count = b.sc; a.total_items = (uint32_t)(f.Q * b.sc);
    a.samples = ctx->d_samples; a.ray_counter = ctx->d_rays; a.work_counter = ctx->d_work;
    a.stacks = ctx->d_stacks; a.stack_stride = (uint32_t)ctx->stacks_lanes; a.staging = ctx->d_staging;
    a.tile_order = t.ordered ? ctx->d_tile_order : nullptr;
    a.tile_lists = ctx->primary_lists ? ctx->d_tile_lists : nullptr;
    a.queue = t.queue ? ctx->d_queue : nullptr;
    a.seg_count = nullptr; a.n_segments = 0;
    a.bvh.nodes = ctx->d_bvh_nodes; a.bvh.leaf_index = ctx->d_bvh_leaf; a.bvh.big = ctx->d_bvh_big; a.bvh.n_big = ctx->bvh_n_big; a.bvh.geom = ctx->d_geom;
    // Chunk size.  A chunk costs its wave a fixed amount (the atomic, and the wave's live paths standing still while it stages the
    // chunk's camera rays), and what a wave still holds when the counter runs dry is proportional to the chunk: the sum is smallest
    // for a chunk proportional to the square root of the items per resident wave.  Measured on the final scene at 1200x800 (ms per
    // image at 31 / 62 / 125 / 250 / 500 samples): 2 sqrt(items per wave) 5.09 / 9.64 / 18.70 / 36.67 / 72.65, flat between factors
    // 1.5 and 3; the earlier rule (items per wave / 16, at most 512) 5.6-5.8 / 10.2-10.8 / 19.1 / 37.0 / 72.7 -- an eighth of config 3,
    // one rank's share on eight GPUs, is where the difference is largest.
    const uint64_t resident_waves = (uint64_t)ctx->num_cu * t.blocks_per_cu * (t.block / 64);
    uint64_t chunk = (uint64_t)((double)(ctx->chunk_factor > 0.0f ? ctx->chunk_factor : 2.0f) * sqrt((double)a.total_items / (double)resident_waves)) / 64 * 64;
    const uint64_t floor_items = (32768ull / (uint64_t)ctx->n_pad + 63) / 64 * 64;     // tiny scenes: keep per-chunk work above the hand-out overhead
    if (chunk < floor_items) chunk = floor_items;
    const uint64_t chunk_cap = ctx->chunk_cap > 0 && ctx->chunk_cap < kChunkItems ? (uint64_t)ctx->chunk_cap : (uint64_t)kChunkItems;
    a.chunk_items = (int)(chunk < 64 ? 64 : (chunk > chunk_cap ? chunk_cap : chunk));
    a.resident_waves = (uint32_t)resident_waves;
    { const uint64_t m = floor_items < 64 ? 64 : floor_items; a.min_chunk_items = (uint32_t)(m < (uint64_t)a.chunk_items ? m : (uint64_t)a.chunk_items); }
    const uint64_t want_blocks = ((uint64_t)a.total_items + (uint64_t)a.chunk_items * (t.block / 64) - 1) / ((uint64_t)a.chunk_items * (t.block / 64));
    const uint64_t cap_blocks = (uint64_t)ctx->num_cu * t.blocks_per_cu;
    const dim3 grid((unsigned)(want_blocks < cap_blocks ? want_blocks : cap_blocks)), block(t.block);
    // the first chunk (or segment) of every launched wave is implied by its index; the counter hands out what follows
    HIP_TRY(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ctx->d_work), (int)(grid.x * (unsigned)(t.block / 64) * (t.queue == 2 ? 1u : (unsigned)a.chunk_items)), 1, f.st));
    if (t.queue == 1) {
        const uint32_t one_each = (a.total_items + 255u) / 256u, shared = (one_each + kPrimaryItemsPerThread - 1u) / kPrimaryItemsPerThread;
        const uint32_t floor_blocks = one_each < kPrimaryMinBlocks ? one_each : kPrimaryMinBlocks;
        hipLaunchKernelGGL(k_primary, dim3(shared > floor_blocks ? shared : floor_blocks), dim3(256), 0, f.st, a);
    }
    if (t.queue == 2) {
        a.n_segments = (a.total_items + (uint32_t)a.chunk_items - 1u) / (uint32_t)a.chunk_items;
        if (ctx->seg_cap < a.n_segments) {
            HIP_TRY(ctx, hipStreamSynchronize(f.st));
            (void)hipFree(ctx->d_seg_count); ctx->d_seg_count = nullptr; ctx->seg_cap = 0;
            HIP_TRY(ctx, hipMalloc(&ctx->d_seg_count, (size_t)a.n_segments * sizeof(uint32_t))); ctx->seg_cap = a.n_segments;
        }
        a.seg_count = ctx->d_seg_count;
        const uint32_t lists_blocks = (a.n_segments + 3u) / 4u, lists_cap = (uint32_t)ctx->num_cu * 8u;
        hipLaunchKernelGGL(k_primary_lists, dim3(lists_blocks < lists_cap ? lists_blocks : lists_cap), dim3(256), 0, f.st, a);
    }
    hipLaunchKernelGGL(trace_kernel(t), grid, block, t.shmem, f.st, a);
    HIP_TRY(ctx, hipGetLastError());
    return RT_OK;
}

int resolve_batch(rt_context* ctx, const RenderFrame& f, const SampleBatch& b)
{
    if (f.tile_output) {
        ResolveTilesArgs r;
        r.samples = ctx->d_samples; r.accum = ctx->d_accum; r.Q = (uint32_t)f.Q; r.tiles_x = f.tiles_x; r.s_count = b.sc;
        r.first = (b.index == 0); r.last = (b.index == f.batches - 1); r.width = f.p->width; r.height = f.p->height; r.spp = f.p->spp;
        r.share_first = f.share_first; r.share_step = f.share_step; r.tile_order = ctx->tile_ordering ? ctx->d_tile_order : nullptr;
        r.rgb_mean = f.d_rgb_mean; r.irgb = f.d_irgb; r.tile_ids = f.d_tile_ids; r.pixel_counter = ctx->d_rays + 2;
        hipLaunchKernelGGL(k_resolve_tiles, dim3((unsigned)((f.Q + 255) / 256)), dim3(256), 0, f.st, r);
        HIP_TRY(ctx, hipGetLastError());
        return RT_OK;
    }
    ResolveArgs r;
    r.samples = ctx->d_samples; r.accum = ctx->d_accum; r.Q = (uint32_t)f.Q; r.tiles_x = f.tiles_x; r.s_count = b.sc;
    r.first = (b.index == 0); r.last = (b.index == f.batches - 1); r.width = f.p->width; r.owned_rows = f.owned; r.spp = f.p->spp;
    r.rgb_mean = f.d_rgb_mean; r.irgb = f.d_irgb;
    // (the wavefront mode fills the sample buffer in image order, the megakernel in its work order)
    r.tile_rank = f.p->mode != RT_MODE_WAVEFRONT && ctx->tile_ordering ? ctx->d_tile_rank : nullptr;
    hipLaunchKernelGGL(k_resolve, dim3((unsigned)((f.Q + 255) / 256)), dim3(256), 0, f.st, r);
    HIP_TRY(ctx, hipGetLastError());
    return RT_OK;
}
