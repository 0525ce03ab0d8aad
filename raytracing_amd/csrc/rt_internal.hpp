// Shared by the translation units of librt_amd.so: the context behind the opaque rt_context handle of
// include/rt_amd.h, error reporting, and a scoped device buffer for the entry points that take host arrays.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "rt_amd.h"
#include "rt_device.hpp"

using namespace rt;

// aux[i] = (radius, type): read once per hit (normal = (p - c) / radius, Scatter dispatch)
struct Aux { float radius; int type; };

struct rt_context {
    int device = -1;
    std::string error;
    float4* d_geom = nullptr; float4* d_mat = nullptr; Aux* d_aux = nullptr; int n = 0;
    float4* d_frec = nullptr; float4* d_prec = nullptr; uint32_t* d_force = nullptr; int n_forced = 0; float plane_s0 = 0.0f; float plane_rmin = 1.0f; int n_pad = 0;
    int chunk_cap = 0; float chunk_factor = 0.0f;                // (RT_AMD_CHUNK_ITEMS, RT_AMD_CHUNK_FACTOR: overrides of the work hand-out's chunk size; measurements)
    bool plane_filter = true; int max_lds_spheres = 0;          // (RT_AMD_MAX_LDS_SPHERES: scenes above it keep their records out of LDS; measurements)
    uint2* d_arec = nullptr; float mfma_g = 0.0f; bool mfma_filter = true;     // matrix-pipe form of the plane filter (closest_hit in rt_sweep.hpp)
    float* d_samples = nullptr; size_t samples_bytes = 0;      // sample buffer of the persistent megakernel
    uint64_t sample_buffer_cap = 16ull << 30;                  // largest sample buffer per batch (bytes)
    float* d_accum = nullptr; size_t accum_bytes = 0;          // running per-pixel sums across sample batches
    unsigned int* d_work = nullptr;
    uint16_t* d_stacks = nullptr; size_t stacks_lanes = 0;
    float* d_staging = nullptr;
    // camera rays of one sample batch as k_primary leaves them for k_trace (32 bytes per item) when camera rays are swept; with per-tile
    // lists the second rays k_primary_lists leaves (up to 48 bytes per item) and how many per segment
    float* d_queue = nullptr; size_t queue_bytes = 0; uint64_t queue_cap = 24ull << 30; bool primary_kernel = true;
    uint32_t* d_seg_count = nullptr; size_t seg_cap = 0;
    uint32_t* d_tile_cls = nullptr; uint32_t* d_tile_order = nullptr; uint32_t* d_tile_rank = nullptr; size_t tile_cap = 0;   // work order of the tiles of the current call (position -> image tile) and its inverse
    bool tile_ordering = true;
    uint32_t* d_tile_lists = nullptr; size_t tile_list_cap = 0;    // per tile: the spheres its camera rays can meet (k_tile_lists)
    bool primary_lists = true;
    bool operand_tiles = false;         // scenes too large for LDS: the filter's operands as LDS tiles that a block stages for its twelve waves (RT_AMD_LARGE_SCENES=tiles); default: every wave streams them from L2, which measures 13 % faster on 20 000 spheres
    // the order depends on scene, camera and the rows rendered only: it is kept for repeated calls (new seeds, more samples)
    struct TileKey { uint64_t scene_version, camera_version; int32_t width, height, band_rows, band_first, band_step; float tmin; } tile_key{};
    bool tile_key_valid = false; uint64_t scene_version = 0, camera_version = 0;
    // wavefront mode pool
    uint32_t wf_P = 0; float* wf_ray = nullptr; uint32_t* wf_pixsmp = nullptr; int* wf_depth = nullptr; float* wf_hit_t = nullptr; int* wf_hit_idx = nullptr;
    uint16_t* wf_stacks = nullptr; uint32_t* wf_q[2] = { nullptr, nullptr }; uint32_t* wf_n = nullptr;
    int wf_blocks_per_cu[4] = { 0, 0, 0, 0 };
    int num_cu = 0; int trace_blocks_per_cu[24] = {};
    // sphere hierarchy of RT_MODE_BVH (rt_bvh.hpp), rebuilt with every scene upload
    float4* d_bvh_nodes = nullptr; uint32_t* d_bvh_leaf = nullptr; uint32_t* d_bvh_big = nullptr; int bvh_n_big = 0, bvh_depth = 0, bvh_nodes = 0;
    bool have_camera = false; CameraD cam{};
    unsigned long long* d_rays = nullptr;      // [0] closest-hit queries, [1] those that hit nothing, [2] pixels of a tile share that lie inside the image
    bool last_was_share = false;               // the last render was a tile share: its primary-ray count is d_rays[2] x spp
    int last_spp = 0;
    uint64_t last_samples = 0;                // primary rays of the last render (owned pixels x spp)
    float* d_out_rgb = nullptr; int32_t* d_out_irgb = nullptr; size_t out_rgb_bytes = 0, out_irgb_bytes = 0;   // device side of rt_render's host outputs (grow-only)
    void* d_ppm_scratch = nullptr; size_t ppm_scratch_bytes = 0;    // per-block lengths and offsets of the PPM encoder
    hipEvent_t ev_ppm = nullptr; bool ppm_pending = false;          // recorded behind the write pass of the last rt_encode_ppm_device
    void* d_ppm_static = nullptr;                                  // this GPU's copy of the static scratch of rt_encode_ppm_device_noalloc (never freed)
    hipEvent_t ev0 = nullptr, ev1 = nullptr; hipStream_t last_stream = nullptr; bool pending = false;
};

// Records the message on the context (or, without one, in a per-thread slot read by rt_last_error(NULL)) and returns `code`.
int fail(rt_context* ctx, int code, const char* fmt, ...) __attribute__((format(printf, 3, 4), visibility("hidden")));
#define HIP_TRY(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(ctx, RT_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); } while (0)

// One rt_render_device call as the kernels' launchers see it.  The image is numbered in 8x8 pixel tiles; Q is the padded
// pixel count (tiles x 64); samples are rendered in `batches` batches of at most s_per samples per pixel.
struct RenderFrame {
    const rt_params* p; int owned;
    int tiles_x; uint64_t Q; int batches, s_per;
    float* d_rgb_mean; int32_t* d_irgb; hipStream_t st;
    // tile share (rt_render_tiles_device): the call renders every share_step-th tile of the work order from share_first on, Q
    // counts its tiles only, and the outputs are tile-major with the image tile of each in d_tile_ids
    uint32_t image_tiles; int share_first, share_step; bool tile_output; uint32_t* d_tile_ids;
};
struct SampleBatch { int index, s0, sc; };
// What trace_prepare settles once per call for the batches of k_trace.
struct TracePlan { bool lds, plane, bvh, mfma, ordered, tiled; int queue; size_t shmem; int blocks_per_cu, block; };

#define RT_HIDDEN __attribute__((visibility("hidden")))
int pixel_render(rt_context* ctx, const RenderFrame& f) RT_HIDDEN;                                              // rt_pixel.hip
int trace_prepare(rt_context* ctx, const RenderFrame& f, TracePlan* plan) RT_HIDDEN;                            // rt_trace.hip
int trace_batch(rt_context* ctx, const RenderFrame& f, const TracePlan& plan, const SampleBatch& b) RT_HIDDEN;  // rt_trace.hip
int resolve_batch(rt_context* ctx, const RenderFrame& f, const SampleBatch& b) RT_HIDDEN;                       // rt_trace.hip
int ppm_static_scratch(rt_context* ctx) RT_HIDDEN;                                                              // rt_ppm.hip
int wavefront_batch(rt_context* ctx, const RenderFrame& f, const SampleBatch& b) RT_HIDDEN;                     // rt_wavefront.hip

inline CameraD to_device(const rt_camera_pod& c)
{
    CameraD d;
    d.origin = { c.origin[0], c.origin[1], c.origin[2] };
    d.llc = { c.lower_left_corner[0], c.lower_left_corner[1], c.lower_left_corner[2] };
    d.horiz = { c.horizontal[0], c.horizontal[1], c.horizontal[2] };
    d.vert = { c.vertical[0], c.vertical[1], c.vertical[2] };
    d.u = { c.u[0], c.u[1], c.u[2] };
    d.v = { c.v[0], c.v[1], c.v[2] };
    d.lens_radius = c.lens_radius;
    return d;
}

inline bool params_ok(const rt_params* p)
{
    return p && p->width > 0 && p->height > 0 && p->spp > 0 && p->max_depth >= 0 && p->max_depth <= RT_MAX_DEPTH &&
           p->band_rows > 0 && p->band_first >= 0 && p->band_step > 0 &&
           (p->mode == RT_MODE_MEGAKERNEL || p->mode == RT_MODE_WAVEFRONT || p->mode == RT_MODE_PIXEL || p->mode == RT_MODE_BVH);
}

namespace {
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t up(const void* h, size_t bytes) { hipError_t e = hipMalloc(&p, bytes ? bytes : 4); if (e == hipSuccess && h && bytes) e = hipMemcpy(p, h, bytes, hipMemcpyHostToDevice); return e; }
    hipError_t down(void* h, size_t bytes) { return hipMemcpy(h, p, bytes, hipMemcpyDeviceToHost); }
    template <class T> T* as() { return static_cast<T*>(p); }
};
}

