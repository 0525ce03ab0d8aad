// RT_MODE_WAVEFRONT: the path tracer as ray queues in HBM.
#include "rt_sweep.hpp"

// ============================================================================ wavefront mode
// The same path tracer as three kernels over ray queues in HBM (mode RT_MODE_WAVEFRONT).  A pool of P path
// slots holds one ray each; a queue lists the live slots.  Per bounce: k_wf_intersect runs the closest-hit
// sweep for every queued ray (all lanes busy, nothing else in the kernel); k_wf_shade scatters or finishes
// each ray, gives a finished slot the next (pixel, sample) item, and writes the surviving slot ids to the
// next queue compacted with a wave ballot + prefix count and one atomic per block.
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
    float* s_disck = reinterpret_cast<float*>(smem + (kLds ? a.sw.n : 0));
    if (kLds) { for (int i = threadIdx.x; i < a.sw.n; i += kTraceBlock) { s_geom[i] = a.sw.geom[i]; s_disck[i] = a.sw.frec[i].w; } }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t* mask_words = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(smem) + (kLds ? ((size_t)a.sw.n * 20 + 15) / 16 * 16 : 0)) + (size_t)wave * kSuperChunkWords * 64 + lane;
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
        if (a.pixsmp[s] != 0xffffffffu) closest_hit<kLds, kPlane, false>(ray, a.tmin, a.sw, ExactLds{ s_geom, nullptr, s_disck }, mask_words, closest, hit, T);
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
                                     a.sw.geom, a.mat, a.aux, StridedStack{ a.stacks + s, (size_t)a.P }, sample, T);
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

// One sample batch: fills the pool, then alternates intersect and shade passes between the two queues until no slot is
// live.  The host steers the loop and reads the live count every 8 bounces (the call blocks on the stream there).
int wavefront_batch(rt_context* ctx, const RenderFrame& f, const SampleBatch& b)
{
    const rt_params* p = f.p;
    hipStream_t st = f.st;
    const uint32_t total = (uint32_t)(f.Q * b.sc);
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
    w.sw.frec = ctx->d_frec; w.sw.prec = ctx->d_prec; w.sw.force = ctx->d_force; w.sw.n_forced = ctx->n_forced; w.sw.plane_s0 = ctx->plane_s0; w.sw.plane_rmin = ctx->plane_rmin; w.sw.geom = ctx->d_geom; w.sw.n = ctx->n; w.sw.n_pad = ctx->n_pad;
    w.mat = ctx->d_mat; w.aux = ctx->d_aux; w.cam = ctx->cam;
    w.ig.width = p->width; w.ig.height = p->height; w.ig.band_rows = p->band_rows; w.ig.band_first = p->band_first; w.ig.band_step = p->band_step;
    w.ig.owned_rows = f.owned; w.ig.tiles_x = f.tiles_x; w.ig.Q = (uint32_t)f.Q; w.ig.s_begin = b.s0; w.ig.s_count = b.sc; w.ig.tile_order = nullptr;
    w.max_depth = p->max_depth; w.tmin = p->tmin; w.k0 = (uint32_t)p->seed; w.k1 = (uint32_t)(p->seed >> 32);
    w.total_items = total; w.P = ctx->wf_P;
    w.ray = ctx->wf_ray; w.pixsmp = ctx->wf_pixsmp; w.depth = ctx->wf_depth; w.hit_t = ctx->wf_hit_t; w.hit_idx = ctx->wf_hit_idx; w.stacks = ctx->wf_stacks;
    w.work_counter = ctx->d_work; w.ray_counter = ctx->d_rays; w.samples = ctx->d_samples;
    const bool wl = ctx->n <= 4096, wp = ctx->plane_filter;      // intersect keeps only the 16-byte exact records in LDS
    const size_t wshmem = (wl ? ((size_t)ctx->n * 20 + 15) / 16 * 16 : 0) + (size_t)(kTraceBlock / 64) * kSuperChunkWords * 64 * sizeof(uint32_t);
    int& wbpc = ctx->wf_blocks_per_cu[(wl ? 1 : 0) + (wp ? 2 : 0)];
    auto ikern = wl ? (wp ? k_wf_intersect<true, true> : k_wf_intersect<true, false>) : (wp ? k_wf_intersect<false, true> : k_wf_intersect<false, false>);
    if (wbpc == 0) {
        int nb = 0;
        const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ikern, kTraceBlock, wshmem);
        if (e != hipSuccess) return fail(ctx, RT_ERR_HIP, "occupancy query for the intersect kernel (%zu bytes of LDS per block) failed: %s", wshmem, hipGetErrorString(e));
        if (nb <= 0) return fail(ctx, RT_ERR_TOO_LARGE, "the intersect kernel does not fit a compute unit with %zu bytes of LDS per block (%d spheres)", wshmem, ctx->n);
        wbpc = nb;
    }
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
        if ((iter & 7) == 7) {
            uint32_t live = 0;
            HIP_TRY(ctx, hipMemcpyAsync(&live, ctx->wf_n + cur, sizeof live, hipMemcpyDeviceToHost, st));
            HIP_TRY(ctx, hipStreamSynchronize(st));
            if (live == 0) break;
            live_upper = live;
        }
    }
    return RT_OK;
}
