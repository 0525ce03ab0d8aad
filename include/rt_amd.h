/* C ABI of the MI355X path-tracing core (librt_amd.so).
 *
 * The reference (IcedLeon/Raytracing) has no plugin or FFI layer; its render loop is the nested
 * loop of main.cpp:118-146, called by main() only.  This library replaces the BODY of that loop:
 * the caller still builds the scene and the camera through the reference's class surface
 * (main.cpp:109-116), flattens them into the plain arrays below, and makes one render call.
 * Each entry point names the reference code it stands in for.
 *
 * Conventions: every function returns 0 on success or a negative rt_status; no exception or
 * abort crosses the boundary; rt_last_error() gives the message of the last failure.  A context
 * is bound to one GPU and is not re-entrant; distinct contexts are independent.
 */
#ifndef RT_AMD_H
#define RT_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_context rt_context;

enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1,      /* bad argument (null pointer, non-positive size, unknown material) */
    RT_ERR_NO_DEVICE = -2,    /* no usable HIP device / ordinal out of range */
    RT_ERR_HIP = -3,          /* a HIP runtime call failed; see rt_last_error */
    RT_ERR_STATE = -4,        /* render requested before a scene and a camera were set */
    RT_ERR_TOO_LARGE = -5     /* size beyond a documented limit */
};

/* Material::Scatter dispatch (material.h:55-59 has no type tag; the flatten pass assigns one). */
enum rt_material { RT_LAMBERTIAN = 0, RT_METAL = 1, RT_DIELECTRIC = 2 };

/* Kernel organisation for the bounce loop of Color() (main.cpp:13-39). */
enum rt_mode {
    RT_MODE_MEGAKERNEL = 0,   /* persistent waves; a lane whose path ends takes the next (pixel, sample)
                                 item at once (ballot + prefix count); conservative FMA filter over the
                                 spheres, exact test on the survivors */
    RT_MODE_WAVEFRONT = 1,    /* a pool of path slots and ray queues in HBM: one intersect and one shade
                                 kernel per bounce, surviving slots compacted with wave ballot/prefix counts;
                                 the host reads the live count every 8 bounces, so the call synchronises */
    RT_MODE_PIXEL = 2,        /* plain kernel: one lane per pixel, literal per-sphere test, scene in LDS */
    RT_MODE_BVH = 3           /* the megakernel with the brute-force sweep of hittablelist.cpp:21-29 replaced by a walk
                                 through a hierarchy of sphere-centre boxes built at upload; same (sphere, t) for every
                                 ray, hence the same image.  Not the path the headline benchmark measures */
};

/* The eight public fields of Camera (camera.h:13-21), copied one to one. */
typedef struct rt_camera_pod {
    float origin[3];
    float lower_left_corner[3];
    float horizontal[3];
    float vertical[3];
    float u[3], v[3], w[3];
    float lens_radius;
} rt_camera_pod;

/* Parameters the reference hard-codes as locals: nx, ny, ns (main.cpp:87-89), the depth cap 50
 * (main.cpp:22) and the hit interval lower bound 0.001 (main.cpp:17).
 *
 * Rows are image rows y counted from the TOP (PPM order, main.cpp:118 walks j = ny-1-y).  A call
 * renders the bands b = band_first, band_first + band_step, ... where band b covers rows
 * [b*band_rows, min((b+1)*band_rows, height)); its output holds exactly those rows, packed in
 * increasing y.  band_first = 0, band_step = 1 renders the whole image.  Because every uniform is
 * a pure function of (seed, pixel, sample, bounce), the assembled image does not depend on how
 * the bands are split over GPUs. */
typedef struct rt_params {
    int32_t width, height;
    int32_t spp;
    int32_t max_depth;        /* 50 in the reference; at most RT_MAX_DEPTH */
    float   tmin;             /* 0.001f in the reference */
    uint64_t seed;            /* Philox key */
    int32_t band_rows, band_first, band_step;
    int32_t mode;             /* enum rt_mode */
} rt_params;

#define RT_MAX_DEPTH 64

/* ---- lifetime ------------------------------------------------------------------------- */
int rt_device_count(int *count);
int rt_create(int device_ordinal, rt_context **out);
void rt_destroy(rt_context *ctx);
/* Message of the most recent failure on ctx (or of the most recent context-less failure when
 * ctx is NULL).  The pointer stays valid until the next failing call. */
const char *rt_last_error(const rt_context *ctx);

/* ---- scene and camera ----------------------------------------------------------------- */
/* Replaces walking HittableList::m_oList (hittablelist.h:10-15) of Sphere{m_oCenter, m_fRadius,
 * m_oMaterial} (sphere.h:10-17) at render time: the caller hands over the flattened list once.
 * Host pointers, n entries each; albedo_rgb holds 3*n floats (Lambertian::m_oAlbedo /
 * Metal::m_oAlbedo; ignored for dielectrics); param is Metal::m_fFuzz (after the constructor's
 * clamp, material.h:73-83) or Dielectric::m_fRefIdx.  The arrays are copied. */
int rt_upload_scene(rt_context *ctx, int n, const float *cx, const float *cy, const float *cz,
                    const float *radius, const int32_t *material, const float *albedo_rgb,
                    const float *param);
/* Replaces reading the Camera object inside GetRay (camera.cpp:46-53). */
int rt_set_camera(rt_context *ctx, const rt_camera_pod *cam);

/* Camera rays (the first Color() call of a sample, main.cpp:129-133) can take a shortcut that is NOT part of the brute-force
 * loop of hittablelist.cpp:21-29: with `enabled` != 0 the megakernel finds their closest hit over a per-tile list of the
 * spheres the tile's camera rays can reach (built every call, see rt_test_tile_lists) instead of sweeping all spheres.  Same
 * image and counters either way.  A new context starts with the shortcut enabled unless the environment variable
 * RT_AMD_PRIMARY is "sweep"; enabled = 0 makes every ray, camera rays included, go through the sweep over all spheres. */
int rt_set_camera_ray_lists(rt_context *ctx, int enabled);

/* ---- render: the loop body of main.cpp:118-141 ---------------------------------------- */
/* Number of image rows / pixels the call described by p produces (0 on invalid p). */
int64_t rt_owned_rows(const rt_params *p);
int64_t rt_owned_pixels(const rt_params *p);

/* Device-pointer form.  d_rgb_mean: 3 floats per owned pixel, the per-pixel mean `col / ns`
 * before gamma (main.cpp:135); d_irgb: 3 int32 per owned pixel, int(255.99*sqrt(mean))
 * (main.cpp:136-140).  Either may be NULL.  Work is enqueued on `stream` (a hipStream_t; NULL =
 * the default stream) and the call returns without waiting, with two exceptions: a call that needs
 * larger internal buffers than the one before waits for the stream before it reallocates them, and
 * RT_MODE_WAVEFRONT steers its bounce loop from the host (it reads the number of live paths back every
 * eight bounces). */
int rt_render_device(rt_context *ctx, const rt_params *p, float *d_rgb_mean, int32_t *d_irgb,
                     void *stream);
/* Waits for the last rt_render_device on ctx and reports the closest-hit queries it traced (one
 * per Color() invocation, the quantity RenderStats::rays_traced counts, main_modern.cpp:25-28)
 * and the GPU time between the HIP events that bracket its kernels.  Either may be NULL. */
int rt_render_stats(rt_context *ctx, uint64_t *rays, float *kernel_ms);
/* The three counters of RenderStats (performance.h:34-37) with the meaning Color() gives them in
 * main_modern.cpp:23-61: Color() invocations, those whose world->Hit() returned true, and those
 * whose Scatter() returned true below the depth cap.  Waits for the last rt_render_device on ctx. */
typedef struct rt_counters {
    uint64_t rays_traced;
    uint64_t intersection_tests;
    uint64_t material_evaluations;
} rt_counters;
int rt_render_counters(rt_context *ctx, rt_counters *out);
/* Host-pointer convenience form: render, copy back, wait. */
int rt_render(rt_context *ctx, const rt_params *p, float *h_rgb_mean, int32_t *h_irgb,
              uint64_t *rays, float *kernel_ms);

/* A share of the image by TILES rather than rows: the tile decomposition of performance.h:133-155 (create_tiles) /
 * main_modern.cpp:197-219 with 8x8-pixel tiles.  p describes the whole image (band_first 0, band_step 1).  The library orders
 * the tiles of an image by how long their paths can get (tiles that show glass, then metal, then diffuse spheres, then sky);
 * share (share_first, share_step) is every share_step-th tile of that order from position share_first on, so that the
 * share_step shares of an image all hold the same mix of expensive and cheap tiles and every tile stays a compact 8x8 block.
 * Outputs, device pointers, any may be NULL: d_irgb_tiles / d_rgb_mean_tiles hold rt_owned_tiles() x 64 x 3 values,
 * tile-major, pixel l of a tile at row l / 8, column l % 8 of the tile, padding pixels of edge tiles zero; d_tile_ids holds
 * the image tile (row-major over ceil(height / 8) x ceil(width / 8) tiles) of each tile of the share.  RT_MODE_MEGAKERNEL and
 * RT_MODE_BVH only.  Same values as rt_render_device for every pixel.  rt_render_stats / rt_render_counters work as after
 * rt_render_device. */
int64_t rt_owned_tiles(const rt_params *p, int share_first, int share_step);
int rt_render_tiles_device(rt_context *ctx, const rt_params *p, int share_first, int share_step, float *d_rgb_mean_tiles,
                           int32_t *d_irgb_tiles, uint32_t *d_tile_ids, void *stream);

/* One image over several GPUs of the node: the tile decomposition of performance.h:133-155 / main_modern.cpp:197-219 with
 * GPUs in place of pool threads.  ctx[0..n) are contexts (normally one per GPU; several on one GPU also work) that all hold
 * the same scene and camera.  p describes the whole image; its band fields are ignored: context k renders tile share (k, n)
 * (rt_render_tiles_device).  One host thread per context runs the render, copies that context's tiles to the host and puts
 * them into their places in the caller's images (the image is wanted on the host for output.ppm, so a hop through GPU 0 would
 * only add a copy).  h_irgb / h_rgb_mean: whole image, height x width x 3, either may be NULL.  rays: sum over the contexts;
 * kernel_ms: n values, one per context (may be NULL).  The assembled image is the one a single context renders, bit for
 * bit, because every uniform is keyed on the pixel. */
int rt_render_multi(rt_context **ctx, int n, const rt_params *p, float *h_rgb_mean, int32_t *h_irgb, uint64_t *rays,
                    float *kernel_ms);

/* ---- scene setup helpers (host side, untimed) --------------------------------------------- */
/* RandomScene() (main.cpp:41-83) generalised to a (2*half_extent)^2 grid of candidate cells with an
 * explicit, compiler-independent draw order: candidate t = (a + half_extent) * 2*half_extent +
 * (b + half_extent) takes Philox4x32-10 block (t, 0, 0, 0) under key `seed` as (choose_mat,
 * jitter_x, jitter_z, -) and blocks (t, 1, 0, 0), (t, 2, 0, 0) for its material draws, each uniform
 * being (word >> 8) * 2^-24.  Same ground sphere, 0.9 jitter, exclusion test around (4, 0.2, 0),
 * 80 / 15 / 5 % material rule and three big spheres as the reference; at most max_small small
 * spheres are kept (in candidate order).  half_extent = 11 gives the reference's 22 x 22 layout;
 * half_extent = 75, max_small = 19996 gives the 20 000-sphere stress scene.  Writes rows of
 * (cx cy cz radius material albedo_r albedo_g albedo_b param) into table9[cap][9]; *n_out = rows. */
int rt_scene_grid(uint64_t seed, int half_extent, int max_small, float *table9, int cap, int *n_out);

/* The hierarchy of sphere-centre boxes that RT_MODE_BVH walks instead of the loop of hittablelist.cpp:21-29, as
 * rt_upload_scene builds it (host only, no GPU needed).  nodes8: node_cap rows of (cmin.xyz, rmax (rounded up by a margin), cmax.xyz, link);
 * link is an int32 stored in the float's bits: >= 0 inner node with children link and link + 1, < 0 leaf with
 * -(link + 1) = first * 8 + (count - 1) into leaf_index.  big: spheres every ray tests directly (at most 16).
 * leaf_index needs n entries, big 16.  counts = (nodes, leaf entries, big entries, depth). */
int rt_build_hierarchy(int n, const float *cx, const float *cy, const float *cz, const float *radius,
                       float *nodes8, int node_cap, uint32_t *leaf_index, uint32_t *big, int32_t counts[4]);

/* ---- output.ppm (main.cpp:91-97, 143-147) --------------------------------------------- */
/* Formats "P3\n{w} {h}\n255\n" followed by one "r g b\n" line per pixel into buf.  Returns the
 * number of bytes written, or the required capacity (negated) when cap is too small. */
int64_t rt_format_ppm(const int32_t *irgb, int width, int height, char *buf, int64_t cap);
/* Creates/truncates `path` and writes the P3 image. */
int rt_write_ppm(const char *path, const int32_t *irgb, int width, int height);
/* The same image as binary P6 ("P6\n{w} {h}\n255\n" + one byte per channel): a quarter of the bytes of the
 * text form, for callers that do not need the reference's exact file.  Fails with RT_ERR_INVALID when a value
 * is outside 0..255 (the reference's int(255.99*c) is not clamped, main.cpp:138-140; the text form keeps such
 * values, a byte cannot). */
int rt_write_ppm_binary(const char *path, const int32_t *irgb, int width, int height);
/* The same text produced on the GPU from a device-resident quantised image (the d_irgb of
 * rt_render_device), so that the only host traffic is the finished text.  d_text: device buffer of `cap`
 * bytes; *bytes receives the length of the text.  When d_text is NULL or cap is too small nothing is
 * written, *bytes holds the required capacity and RT_ERR_TOO_LARGE is returned.  The call waits for the
 * length pass on `stream`; the write pass is enqueued on it and not waited for. */
int rt_encode_ppm_device(rt_context *ctx, const int32_t *d_irgb, int width, int height, char *d_text,
                         int64_t cap, int64_t *bytes, void *stream);
/* rt_encode_ppm_device for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  It has no counterpart in the reference.  The per-block lengths and offsets live in a
 * fixed array inside the library instead of allocated memory, which limits the image to RT_PPM_NOALLOC_MAX_PIXELS pixels
 * (RT_ERR_TOO_LARGE beyond); the call allocates nothing, copies nothing to the host and does not wait: the text length goes
 * to *d_bytes, an int64 in DEVICE memory, and the text is written only if that length fits `cap` (decided on the GPU;
 * d_text may be NULL to get the length alone).  All work is enqueued on `stream`.  Every call on one GPU uses the same
 * array: calls that overlap in time on one GPU (two streams, two contexts) race and must be serialised by the caller. */
#define RT_PPM_NOALLOC_MAX_PIXELS (16u << 20)
int rt_encode_ppm_device_noalloc(rt_context *ctx, const int32_t *d_irgb, int width, int height, char *d_text,
                                 int64_t cap, int64_t *d_bytes, void *stream);
/* rt_encode_ppm_device into a temporary device buffer, one copy to the host, then the file write. */
int rt_write_ppm_device(rt_context *ctx, const char *path, const int32_t *d_irgb, int width, int height,
                        void *stream);

/* ---- unit-level entry points: the device functions on arrays -------------------------- */
/* Each runs the same __device__ function the render kernels call, one element per lane, on host
 * arrays of `n` elements.  They exist so that Sphere::Hit, the three Scatter functions and
 * Camera::GetRay can be checked in isolation against vectors produced by the reference. */
/* Sphere::Hit (sphere.cpp:15-52). in: ray_o[3n], ray_d[3n], center[3n], radius[n], tmin[n],
 * tmax[n]; out: hit[n], t[n], p[3n], normal[3n] (front_face is dot(d, normal) <= 0, the test
 * Dielectric::Scatter makes at material.h:148). */
int rt_test_sphere_hit(rt_context *ctx, int n, const float *ray_o, const float *ray_d,
                       const float *center, const float *radius, const float *tmin,
                       const float *tmax, int32_t *hit, float *t, float *p, float *normal);
/* Material::Scatter (material.h:85-91, 107-116, 132-186).  draws: draws_per_case rand() integers
 * per case, consumed in call order as rand()/(RAND_MAX+1.0).  out: ret[n], attenuation[3n],
 * out_o[3n], out_d[3n], used[n] (draws consumed). */
int rt_test_scatter(rt_context *ctx, int n, const int32_t *material, const float *albedo_rgb,
                    const float *param, const float *ray_o, const float *ray_d, const float *hit_p,
                    const float *hit_n, const int32_t *draws, int draws_per_case, int32_t *ret,
                    float *attenuation, float *out_o, float *out_d, int32_t *used);
/* Camera::GetRay (camera.cpp:46-53) with an explicit unit-disk sample per case. */
int rt_test_get_ray(rt_context *ctx, int n, const rt_camera_pod *cam, const float *s,
                    const float *t, const float *disk_xy, float *ray_o, float *ray_d);
/* Reflect, Refract and Schlick (material.h:35-38, 16-33, 7-14) on arrays: v[3n], normal[3n], ni_over_nt[n], cosine[n],
 * ref_idx[n] -> reflected[3n], refract_ok[n], refracted[3n] (zeros where Refract fails), schlick[n]. */
int rt_test_helpers(rt_context *ctx, int n, const float *v, const float *normal, const float *ni_over_nt,
                    const float *cosine, const float *ref_idx, float *reflected, int32_t *refract_ok,
                    float *refracted, float *schlick);
/* The candidate lists that the default render path builds for the camera rays of the rows `p` selects, copied to the host: per
 * 8x8 pixel tile (row-major over the owned rows) 64 words, word 0 = number of spheres (0xffffffff: no list, the tile's camera rays
 * are swept over all spheres), words 1.. = their indices in ascending order.  The first level of Color() (main.cpp:13-39) for a
 * camera ray searches only its tile's list, so every sphere HittableList::Hit (hittablelist.cpp:13-31) could report for a ray of the
 * tile (Camera::GetRay, camera.cpp:46-53, over the tile's pixels, all jitters and the whole lens) has to be on it.  *tiles receives
 * the number of tiles; RT_ERR_TOO_LARGE when capacity_words < 64 * *tiles.  The lists live in the context and are rebuilt by every
 * render call: like every call on a context, not to be made while another call on the same context is in flight. */
int rt_test_tile_lists(rt_context *ctx, const rt_params *p, uint32_t *lists, int64_t capacity_words, int64_t *tiles);
/* Philox4x32-10 blocks: ctr[4n], key[2] -> out[4n]. */
int rt_test_philox(rt_context *ctx, int n, const uint32_t *ctr, const uint32_t *key, uint32_t *out);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
