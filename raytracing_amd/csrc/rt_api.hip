// The extern "C" layer of librt_amd.so for contexts, scenes, cameras and renders (include/rt_amd.h).  The kernels and
// their launchers are in rt_pixel.hip, rt_trace.hip and rt_wavefront.hip.
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>
#include "rt_internal.hpp"
#include "rt_bvh.hpp"
static constexpr int kMaxLdsSpheresLimit = 1536;      // kMaxLdsSpheres of rt_sweep.hpp (this file does not include the kernels' header)

static thread_local std::string g_error;

int fail(rt_context* ctx, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (ctx) ctx->error = buf; else g_error = buf;
    return code;
}

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
    // RT_AMD_FILTER=plane keeps the plane filter on the vector ALU (records as scalar operands) instead of the matrix pipe.
    { const char* f = getenv("RT_AMD_FILTER"); ctx->plane_filter = !(f && strcmp(f, "disc") == 0); ctx->mfma_filter = !(f && (strcmp(f, "disc") == 0 || strcmp(f, "plane") == 0)); }
    // RT_AMD_TILE_ORDER=image works through the tiles in image order instead of longest-paths-first (A/B runs).
    // RT_AMD_SAMPLE_BUFFER_KB caps the sample buffer (tests use it to run the multi-batch path on small images).
    { const char* f = getenv("RT_AMD_SAMPLE_BUFFER_KB"); const long long kb = f ? atoll(f) : 0; if (kb > 0) ctx->sample_buffer_cap = (uint64_t)kb << 10; }
    { const char* f = getenv("RT_AMD_MAX_LDS_SPHERES"); const int v = f ? atoi(f) : 0; if (v > 0 && v <= kMaxLdsSpheresLimit) ctx->max_lds_spheres = v; }
    { const char* f = getenv("RT_AMD_CHUNK_ITEMS"); const int v = f ? atoi(f) : 0; if (v >= 64) ctx->chunk_cap = v / 64 * 64; }
    { const char* f = getenv("RT_AMD_CHUNK_FACTOR"); const double v = f ? atof(f) : 0.0; if (v > 0.0) ctx->chunk_factor = (float)v; }
    // RT_AMD_PRIMARY_KERNEL=0: swept camera rays are generated inside the render kernel, chunk by chunk, instead of by a kernel of
    // their own (A/B runs); RT_AMD_QUEUE_KB caps that kernel's queue (tests use it to run the multi-batch path on small images).
    { const char* f = getenv("RT_AMD_PRIMARY_KERNEL"); ctx->primary_kernel = !(f && strcmp(f, "0") == 0); }
    { const char* f = getenv("RT_AMD_QUEUE_KB"); const long long kb = f ? atoll(f) : 0; if (kb > 0) ctx->queue_cap = (uint64_t)kb << 10; }
    { const char* f = getenv("RT_AMD_TILE_ORDER"); ctx->tile_ordering = !(f && strcmp(f, "image") == 0); }
    { const char* f = getenv("RT_AMD_PRIMARY"); ctx->primary_lists = !(f && strcmp(f, "sweep") == 0); }
    { const char* f = getenv("RT_AMD_LARGE_SCENES"); ctx->operand_tiles = f && strcmp(f, "tiles") == 0; }
    hipError_t e = hipSetDevice(device_ordinal);
    if (e == hipSuccess) e = hipMalloc(&ctx->d_rays, 3 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc(&ctx->d_work, sizeof(unsigned int));
    if (e == hipSuccess) { hipDeviceProp_t prop; e = hipGetDeviceProperties(&prop, device_ordinal); if (e == hipSuccess) ctx->num_cu = prop.multiProcessorCount; }
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev0);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev1);
    if (e != hipSuccess) { fail(nullptr, RT_ERR_HIP, "rt_create: %s", hipGetErrorString(e)); rt_destroy(ctx); return RT_ERR_HIP; }
    (void)ppm_static_scratch(ctx);      // a failure here only disables rt_encode_ppm_device_noalloc, which then reports it
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
    (void)hipFree(ctx->d_frec); (void)hipFree(ctx->d_prec); (void)hipFree(ctx->d_arec); (void)hipFree(ctx->d_force); (void)hipFree(ctx->d_samples); (void)hipFree(ctx->d_accum); (void)hipFree(ctx->d_work); (void)hipFree(ctx->d_stacks); (void)hipFree(ctx->d_staging); (void)hipFree(ctx->d_tile_cls); (void)hipFree(ctx->d_tile_order); (void)hipFree(ctx->d_tile_rank); (void)hipFree(ctx->d_queue); (void)hipFree(ctx->d_seg_count); (void)hipFree(ctx->d_tile_lists);
    (void)hipFree(ctx->wf_ray); (void)hipFree(ctx->wf_pixsmp); (void)hipFree(ctx->wf_depth); (void)hipFree(ctx->wf_hit_t); (void)hipFree(ctx->wf_hit_idx);
    (void)hipFree(ctx->wf_stacks); (void)hipFree(ctx->wf_q[0]); (void)hipFree(ctx->wf_q[1]); (void)hipFree(ctx->wf_n);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->ev_ppm) (void)hipEventDestroy(ctx->ev_ppm);
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
    // Plane-filter records (cx, cz, 2^88 R_i, 0), R_i = r^2 + slack_i.  u = 2^-24.  Allowance for the reference's own
    // discriminant rounding: 32 u (|o-c|^2 + r^2) <= 64 u (|o|^2 + |c|^2) + 32 u r^2; this form's rounding
    // of pu (< 8 u (|c| + |o|)) moves pu^2 near the threshold by < 16 u |r| (|c| + |o|)
    // <= 8 u (2 r^2 + |c|^2 + |o|^2); the normalised form of a chunk (below) adds the roundings of cx / rho, cz / rho
    // and (o.u) / rho, another 5 u (|c| + |o|) on pu.  slack_i = u (144 |c|^2 + 80 r^2), S_ray = 144 u |o|^2 (kernel).
    // The kernel reads the verdict of a pair as the number 0.0 or 1.0, which it is only for 2^-38 <= R_i <= 2^38 (see
    // filter_bits in rt_sweep.hpp); spheres outside that range -- zero-radius ones included -- get a record that never
    // passes and a bit in `force`: they always go on to the next test.  Padding records never pass.
    std::vector<float4> prec(n_pad, make_float4(0.0f, 0.0f, -FLT_MAX, 0.0f));
    const int n_words = (n_pad + 31) / 32;
    std::vector<uint32_t> force(n_words + (n_words + 15) / 16, 0u); int n_forced = 0;      // forced bits, then one "stored normalised" bit per chunk
    std::vector<float> Rof(n_pad, 0.0f); std::vector<char> judged_flag(n_pad, 0);
    double rmin_R = 1e30;
    for (int i = 0; i < n; ++i) {
        const double rr = (double)radius[i] * radius[i];
        const double cc = (double)cx[i] * cx[i] + (double)cy[i] * cy[i] + (double)cz[i] * cz[i];
        const double R = rr + 5.9604644775390625e-08 * (144.0 * cc + 80.0 * rr) + 1e-30;
        float rf = (float)R; if ((double)rf < R) rf = nextafterf(rf, INFINITY);
        const bool judged = rf >= 0x1p-38f && rf <= 0x1p38f && std::isfinite(cx[i]) && std::isfinite(cz[i]);
        Rof[i] = rf; judged_flag[i] = judged ? 1 : 0;
        if (judged) {
            prec[i] = make_float4(cx[i], cz[i], ldexpf(rf, 88), 0.0f);
            if ((double)rf < rmin_R) rmin_R = (double)rf;
        } else {
            force[i >> 5] |= 0x80000000u >> (i & 31); ++n_forced;
        }
    }
    // Full chunks of 32 judged spheres whose R lie within 20 % of each other are stored normalised by rho = sqrt(max R)
    // (rounded up), see filter_norm in rt_sweep.hpp.
    for (int base = 0; base + 32 <= n_pad && base + 32 <= n; base += 32) {
        float rmax = 0.0f, rmin = FLT_MAX; bool all = true;
        for (int i = base; i < base + 32; ++i) { all = all && judged_flag[i]; rmax = std::max(rmax, Rof[i]); rmin = std::min(rmin, Rof[i]); }
        if (!all || rmin < 0.8f * rmax) continue;
        const double rho = std::sqrt((double)rmax) * (1.0 + 1e-7);
        // compact layout inside the chunk's 32 float4 slots: 32 pairs (cx / rho, cz / rho) in the first 64 floats, 1 / rho in float 64,
        // so that one 16-dword scalar load brings eight spheres and half as many bytes stream through the scalar cache
        float* slot = reinterpret_cast<float*>(&prec[base]);
        for (int k = 0; k < 128; ++k) slot[k] = 0.0f;
        for (int i = 0; i < 32; ++i) { slot[2 * i] = (float)(cx[base + i] / rho); slot[2 * i + 1] = (float)(cz[base + i] / rho); }
        slot[64] = (float)(1.0 / rho);
        force[n_words + (base >> 9)] |= 0x8000u >> ((base >> 5) & 15);
    }
    // The same filter for the matrix pipe (filter_mfma in rt_sweep.hpp): every judged sphere normalised by its own rho, as bf16
    // heads and tails.  q is formed from bf16 pieces whose omitted cross terms and tails are < 2^-14 of the three products'
    // magnitudes (2^-13 is allowed for, which also covers the accumulator's roundings and those of the kernel's float normal):
    //   rho >= sqrt(R) + 2^-13 (|cx| + |cz|), and 1 / rho rounded DOWN to a bf16 number (rho grows by < 2^-7 of itself);
    //   the ray's share, 2^-13 |o.u| <= 2^-13 (|o.x| + |o.z|), shrinks the ray's normal by the factor 1 - (|o.x| + |o.z|) g with
    //   g = 2^-13 / sqrt(min R) (mfma_g), which brings kappa |pu| + 2^-13 |o.u| under sqrt(R) for every judged sphere.
    // Row order inside a chunk: the instruction returns rows 8 (v / 4) + 4 h + v % 4 to lane half h, register v, and the
    // kernel wants those to be spheres 16 h + v.  Spheres that are not judged, and padding, get NaN records: no ray passes.
    const int n_chunks = (n + 31) / 32;
    std::vector<uint16_t> arec((size_t)n_chunks * 64 * 4, (uint16_t)0x7fc0);
    auto bf16_rn = [](double x) -> uint16_t {           // nearest bf16, ties to even; x finite
        const float f = (float)x; uint32_t b; memcpy(&b, &f, 4);
        // a double rounded to float first can round twice; the one extra half-ulp of the float (2^-25 relative) is nothing here
        b += 0x7fffu + ((b >> 16) & 1u);
        return (uint16_t)(b >> 16);
    };
    auto bf16_val = [](uint16_t h) -> double { const uint32_t b = (uint32_t)h << 16; float f; memcpy(&f, &b, 4); return (double)f; };
    for (int i = 0; i < n; ++i) {
        if (!judged_flag[i]) continue;
        const double rho = (std::sqrt((double)Rof[i]) + 0x1p-13 * (std::fabs((double)cx[i]) + std::fabs((double)cz[i]))) * (1.0 + 1e-6);
        const float irf = (float)(1.0 / rho); uint32_t ib; memcpy(&ib, &irf, 4);
        if ((double)irf > 1.0 / rho) --ib;
        const uint16_t ir = (uint16_t)(ib >> 16);                   // truncation of a positive float: rounded down
        const double irv = bf16_val(ir);
        const double ax = (double)cx[i] * irv, az = (double)cz[i] * irv;
        const bool fits = irv >= 0x1p-60 && irv <= 0x1p60 && std::fabs(ax) <= 0x1p60 && std::fabs(az) <= 0x1p60;
        if (!fits) { force[i >> 5] |= 0x80000000u >> (i & 31); ++n_forced; continue; }     // (forced for every filter form: always a candidate)
        const uint16_t axh = bf16_rn(ax), azh = bf16_rn(az);
        const uint16_t axl = bf16_rn(ax - bf16_val(axh)), azl = bf16_rn(az - bf16_val(azh));
        const int c = i >> 5, j = i & 31, h = j >> 4, v = j & 15;
        const int row = 8 * (v >> 2) + 4 * h + (v & 3);
        uint16_t* lo = &arec[((size_t)c * 64 + row) * 4];            // lane `row`: k 0..3
        uint16_t* hi = &arec[((size_t)c * 64 + 32 + row) * 4];       // lane 32 + `row`: k 4..7
        lo[0] = axh; lo[1] = axl; lo[2] = axh; lo[3] = azh;
        hi[0] = azl; hi[1] = azh; hi[2] = ir; hi[3] = ir;
    }
    const float plane_s0 = 1e-30f;
    if (rmin_R > 1e29) rmin_R = 1.0;
    float mfma_g = (float)(0x1p-13 / std::sqrt(rmin_R) * (1.0 + 1e-6)); mfma_g = nextafterf(mfma_g, INFINITY);
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
    (void)hipFree(ctx->d_geom); (void)hipFree(ctx->d_mat); (void)hipFree(ctx->d_aux); (void)hipFree(ctx->d_frec); (void)hipFree(ctx->d_prec); (void)hipFree(ctx->d_arec); (void)hipFree(ctx->d_force);
    ctx->d_geom = nullptr; ctx->d_mat = nullptr; ctx->d_aux = nullptr; ctx->d_frec = nullptr; ctx->d_prec = nullptr; ctx->d_arec = nullptr; ctx->d_force = nullptr; ctx->n = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->d_arec, arec.size() * sizeof(uint16_t)));
    HIP_TRY(ctx, hipMemcpy(ctx->d_arec, arec.data(), arec.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    ctx->mfma_g = mfma_g;
    HIP_TRY(ctx, hipMalloc(&ctx->d_prec, sizeof(float4) * n_pad));
    HIP_TRY(ctx, hipMemcpy(ctx->d_prec, prec.data(), sizeof(float4) * n_pad, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMalloc(&ctx->d_force, sizeof(uint32_t) * force.size()));
    HIP_TRY(ctx, hipMemcpy(ctx->d_force, force.data(), sizeof(uint32_t) * force.size(), hipMemcpyHostToDevice));
    ctx->plane_s0 = plane_s0; ctx->plane_rmin = plane_rmin; ctx->n_forced = n_forced;
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

int rt_set_camera_ray_lists(rt_context* ctx, int enabled)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "rt_set_camera_ray_lists: null context");
    // the tile classes come from the lists when there are lists and from probe rays otherwise: the kept work order is for one of the two
    if (ctx->primary_lists != (enabled != 0)) ctx->tile_key_valid = false;
    ctx->primary_lists = enabled != 0;
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

}  // extern "C"
namespace {
// Tiles of the image described by p (8x8 pixels, edge tiles padded).
int64_t image_tiles(const rt_params* p) { return (int64_t)((p->width + 7) / 8) * ((p->height + 7) / 8); }

// rt_render_device and rt_render_tiles_device.  share_step = 1, share_first = 0, tile_output = false: the rows p selects, row-major.
int render_impl(rt_context* ctx, const char* who, const rt_params* p, int share_first, int share_step, bool tile_output,
                float* d_rgb_mean, int32_t* d_irgb, uint32_t* d_tile_ids, void* stream)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "%s: null context", who);
    if (!params_ok(p)) return fail(ctx, RT_ERR_INVALID, "%s: invalid parameters", who);
    if (!ctx->n || !ctx->have_camera) return fail(ctx, RT_ERR_STATE, "%s: scene and camera must be set first", who);
    if (tile_output) {
        if (share_step <= 0 || share_first < 0 || share_first >= share_step) return fail(ctx, RT_ERR_INVALID, "%s: share %d of %d", who, share_first, share_step);
        if (p->band_first != 0 || p->band_step != 1) return fail(ctx, RT_ERR_INVALID, "%s: a tile share is a share of the whole image (band_first 0, band_step 1)", who);
        if (p->mode != RT_MODE_MEGAKERNEL && p->mode != RT_MODE_BVH) return fail(ctx, RT_ERR_INVALID, "%s: tile shares are rendered by the megakernel (RT_MODE_MEGAKERNEL, RT_MODE_BVH)", who);
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int owned = (int)rt_owned_rows(p);
    // The counters, the sample buffer, the staging rows, the hit stacks and the tile order belong to the context, not to the
    // call: a render enqueued on another stream than the one before has to start after that one has finished with them.
    if (ctx->pending && st != ctx->last_stream) HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev1, 0));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_rays, 0, 3 * sizeof(unsigned long long), st));
    HIP_TRY(ctx, hipEventRecord(ctx->ev0, st));
    const int64_t all_tiles = image_tiles(p);
    const int64_t share_tiles = tile_output ? (all_tiles - share_first + share_step - 1) / share_step : 0;
    if (owned > 0 && !(tile_output && share_tiles <= 0)) {
        RenderFrame f{};
        f.p = p; f.owned = owned; f.d_rgb_mean = d_rgb_mean; f.d_irgb = d_irgb; f.st = st;
        f.share_first = 0; f.share_step = 1; f.tile_output = false; f.d_tile_ids = nullptr;
        if (p->mode == RT_MODE_PIXEL) {
            const int rc = pixel_render(ctx, f);
            if (rc != RT_OK) return rc;
        } else {
            f.tiles_x = (p->width + 7) / 8;
            f.image_tiles = (uint32_t)((uint64_t)f.tiles_x * ((owned + 7) / 8));
            f.Q = (uint64_t)f.image_tiles * 64;
            if (tile_output) { f.share_first = share_first; f.share_step = share_step; f.tile_output = true; f.d_tile_ids = d_tile_ids; f.Q = (uint64_t)share_tiles * 64; }
            // sample batches: the per-batch sample buffer stays below 16 GiB (of 288 GB of HBM) and item ids below 2^31;
            // every batch ends with the drain of its longest paths, so fewer, larger batches are better
            uint64_t per_batch = ctx->sample_buffer_cap / (f.Q * 12);
            // (with the camera rays of a batch queued by a kernel of their own -- 32 or 48 bytes per item -- that queue has a cap too)
            const uint64_t row_bytes = ctx->primary_lists ? 48 : 32;
            if (p->mode != RT_MODE_WAVEFRONT && ctx->primary_kernel && per_batch > ctx->queue_cap / (f.Q * row_bytes)) per_batch = ctx->queue_cap / (f.Q * row_bytes);
            if (per_batch > (1ull << 31) / f.Q) per_batch = (1ull << 31) / f.Q;
            if (per_batch == 0) return fail(ctx, RT_ERR_TOO_LARGE, "%s: %llu pixels in one call exceed the sample-buffer (or camera-ray queue) limit; split the image into bands", who, (unsigned long long)f.Q);
            f.batches = (int)((p->spp + per_batch - 1) / per_batch);
            f.s_per = (p->spp + f.batches - 1) / f.batches;
            const size_t need = (size_t)f.Q * f.s_per * 12;
            if (ctx->samples_bytes < need) {
                HIP_TRY(ctx, hipStreamSynchronize(st));
                (void)hipFree(ctx->d_samples); ctx->d_samples = nullptr; ctx->samples_bytes = 0;
                HIP_TRY(ctx, hipMalloc(&ctx->d_samples, need)); ctx->samples_bytes = need;
            }
            if (f.batches > 1 && ctx->accum_bytes < f.Q * 12) {
                HIP_TRY(ctx, hipStreamSynchronize(st));
                (void)hipFree(ctx->d_accum); ctx->d_accum = nullptr; ctx->accum_bytes = 0;
                HIP_TRY(ctx, hipMalloc(&ctx->d_accum, f.Q * 12)); ctx->accum_bytes = f.Q * 12;
            }
            const bool wavefront = p->mode == RT_MODE_WAVEFRONT;
            TracePlan plan{};
            if (!wavefront) { const int rc = trace_prepare(ctx, f, &plan); if (rc != RT_OK) return rc; }
            for (int bi = 0, s0 = 0; bi < f.batches; ++bi, s0 += f.s_per) {
                const SampleBatch b{ bi, s0, (p->spp - s0 < f.s_per) ? p->spp - s0 : f.s_per };
                int rc = wavefront ? wavefront_batch(ctx, f, b) : trace_batch(ctx, f, plan, b);
                if (rc == RT_OK) rc = resolve_batch(ctx, f, b);
                if (rc != RT_OK) return rc;
            }
        }
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev1, st));
    ctx->last_stream = st; ctx->pending = true; ctx->last_samples = (uint64_t)rt_owned_pixels(p) * (uint64_t)p->spp;
    ctx->last_was_share = tile_output; ctx->last_spp = p->spp;
    return RT_OK;
}
}  // namespace
extern "C" {

int rt_render_device(rt_context* ctx, const rt_params* p, float* d_rgb_mean, int32_t* d_irgb, void* stream)
{
    return render_impl(ctx, "rt_render_device", p, 0, 1, false, d_rgb_mean, d_irgb, nullptr, stream);
}

int64_t rt_owned_tiles(const rt_params* p, int share_first, int share_step)
{
    if (!params_ok(p) || share_step <= 0 || share_first < 0 || share_first >= share_step) return 0;
    const int64_t all = image_tiles(p);
    return all > share_first ? (all - share_first + share_step - 1) / share_step : 0;
}

int rt_render_tiles_device(rt_context* ctx, const rt_params* p, int share_first, int share_step, float* d_rgb_mean_tiles, int32_t* d_irgb_tiles,
                           uint32_t* d_tile_ids, void* stream)
{
    return render_impl(ctx, "rt_render_tiles_device", p, share_first, share_step, true, d_rgb_mean_tiles, d_irgb_tiles, d_tile_ids, stream);
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
    uint64_t primary = ctx->last_samples;
    if (ctx->last_was_share) {                 // a tile share: its pixels were counted on the device
        unsigned long long px = 0;
        HIP_TRY(ctx, hipMemcpy(&px, ctx->d_rays + 2, sizeof px, hipMemcpyDeviceToHost));
        primary = (uint64_t)px * (uint64_t)ctx->last_spp;
    }
    out->material_evaluations = c[0] - primary;
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
namespace {
// Renders context k's share of the image -- every n-th tile of the work order (most expensive tiles first), so that all contexts
// get the same mix -- and puts its tiles into their places in the caller's whole-image arrays.
int render_share(rt_context* ctx, int k, int n, const rt_params& whole, float* h_rgb_mean, int32_t* h_irgb, uint64_t* rays, float* ms)
{
    rt_params p = whole;
    p.band_first = 0; p.band_step = 1; p.band_rows = whole.height;
    if (p.mode != RT_MODE_MEGAKERNEL && p.mode != RT_MODE_BVH) p.mode = RT_MODE_MEGAKERNEL;      // (the other organisations know no tile shares; same image)
    const int64_t tiles = rt_owned_tiles(&p, k, n);
    *rays = 0; *ms = 0.0f;
    if (tiles == 0) return RT_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t npx = (size_t)tiles * 64;
    // (the device side of the outputs is kept in the context like rt_render's; the tile ids follow the integers or the floats)
    const size_t id_bytes = (size_t)tiles * sizeof(uint32_t);
    if (h_rgb_mean && ctx->out_rgb_bytes < npx * 12 + id_bytes) {
        (void)hipFree(ctx->d_out_rgb); ctx->d_out_rgb = nullptr; ctx->out_rgb_bytes = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_out_rgb, npx * 12 + id_bytes)); ctx->out_rgb_bytes = npx * 12 + id_bytes;
    }
    if (h_irgb && ctx->out_irgb_bytes < npx * 12 + id_bytes) {
        (void)hipFree(ctx->d_out_irgb); ctx->d_out_irgb = nullptr; ctx->out_irgb_bytes = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_out_irgb, npx * 12 + id_bytes)); ctx->out_irgb_bytes = npx * 12 + id_bytes;
    }
    if (!h_rgb_mean && !h_irgb) {
        int rc = rt_render_tiles_device(ctx, &p, k, n, nullptr, nullptr, nullptr, nullptr);
        if (rc == RT_OK) rc = rt_render_stats(ctx, rays, ms);
        return rc;
    }
    uint32_t* d_ids = reinterpret_cast<uint32_t*>(h_irgb ? reinterpret_cast<char*>(ctx->d_out_irgb) + npx * 12 : reinterpret_cast<char*>(ctx->d_out_rgb) + npx * 12);
    int rc = rt_render_tiles_device(ctx, &p, k, n, h_rgb_mean ? ctx->d_out_rgb : nullptr, h_irgb ? ctx->d_out_irgb : nullptr, d_ids, nullptr);
    if (rc == RT_OK) rc = rt_render_stats(ctx, rays, ms);
    if (rc != RT_OK) return rc;
    std::vector<uint32_t> ids((size_t)tiles);
    HIP_TRY(ctx, hipMemcpy(ids.data(), d_ids, id_bytes, hipMemcpyDeviceToHost));
    const int tiles_x = (p.width + 7) / 8;
    auto scatter = [&](void* host, const void* dev) -> int {
        std::vector<uint32_t> packed(npx * 3);           // 12 bytes per pixel, floats or integers alike
        HIP_TRY(ctx, hipMemcpy(packed.data(), dev, npx * 12, hipMemcpyDeviceToHost));
        uint32_t* img = static_cast<uint32_t*>(host);
        for (int64_t t = 0; t < tiles; ++t) {
            const int ty = (int)(ids[(size_t)t] / (uint32_t)tiles_x), tx = (int)(ids[(size_t)t] % (uint32_t)tiles_x);
            const int w = p.width - tx * 8 < 8 ? p.width - tx * 8 : 8, h = p.height - ty * 8 < 8 ? p.height - ty * 8 : 8;
            for (int r = 0; r < h; ++r)
                memcpy(img + ((size_t)(ty * 8 + r) * p.width + (size_t)tx * 8) * 3, packed.data() + ((size_t)t * 64 + (size_t)r * 8) * 3, (size_t)w * 12);
        }
        return RT_OK;
    };
    if (h_rgb_mean) { rc = scatter(h_rgb_mean, ctx->d_out_rgb); if (rc != RT_OK) return rc; }
    if (h_irgb) { rc = scatter(h_irgb, ctx->d_out_irgb); if (rc != RT_OK) return rc; }
    return RT_OK;
}
}  // namespace

extern "C" int rt_render_multi(rt_context** ctx, int n, const rt_params* p, float* h_rgb_mean, int32_t* h_irgb, uint64_t* rays, float* kernel_ms)
{
    if (!ctx || n <= 0) return fail(nullptr, RT_ERR_INVALID, "rt_render_multi: no contexts");
    for (int k = 0; k < n; ++k) if (!ctx[k]) return fail(nullptr, RT_ERR_INVALID, "rt_render_multi: context %d is null", k);
    for (int k = 0; k < n; ++k) for (int j = 0; j < k; ++j) if (ctx[j] == ctx[k]) return fail(ctx[0], RT_ERR_INVALID, "rt_render_multi: context %d is listed twice", k);
    if (!p) return fail(ctx[0], RT_ERR_INVALID, "rt_render_multi: invalid parameters");
    rt_params whole = *p;                   // band_first / band_step are documented as ignored: whatever they hold is not judged
    whole.band_first = 0; whole.band_step = n;
    if (!params_ok(&whole)) return fail(ctx[0], RT_ERR_INVALID, "rt_render_multi: invalid parameters");
    for (int k = 0; k < n; ++k) if (!ctx[k]->n || !ctx[k]->have_camera) return fail(ctx[0], RT_ERR_STATE, "rt_render_multi: context %d has no scene or camera", k);
    std::vector<int> rc(n, RT_OK); std::vector<uint64_t> r(n, 0); std::vector<float> ms(n, 0.0f);
    std::vector<std::thread> workers;
    // one host thread per context; HIP's current device is per thread
    // (no exception may cross the C boundary: a thread that cannot be started ends the call with an error once the others have joined)
    bool started_all = true;
    try {
        workers.reserve((size_t)n);
        for (int k = 1; k < n; ++k) workers.emplace_back([&, k] { rc[k] = render_share(ctx[k], k, n, whole, h_rgb_mean, h_irgb, &r[k], &ms[k]); });
    } catch (const std::exception&) { started_all = false; }
    if (started_all) rc[0] = render_share(ctx[0], 0, n, whole, h_rgb_mean, h_irgb, &r[0], &ms[0]);
    for (std::thread& t : workers) t.join();
    if (!started_all) return fail(ctx[0], RT_ERR_STATE, "rt_render_multi: could not start %d host threads", n - 1);
    uint64_t total = 0;
    for (int k = 0; k < n; ++k) {
        if (rc[k] != RT_OK) { if (k != 0) ctx[0]->error = "context " + std::to_string(k) + ": " + ctx[k]->error; return rc[k]; }
        total += r[k];
        if (kernel_ms) kernel_ms[k] = ms[k];
    }
    if (rays) *rays = total;
    return RT_OK;
}
