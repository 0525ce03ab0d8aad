// RT_MODE_PIXEL: one lane per pixel, the first correct kernel, kept as an independent second implementation.
#include "rt_sweep.hpp"

struct RenderArgs {
    const float4* geom; const float4* mat; const Aux* aux; int n;
    CameraD cam;
    int width, height, spp, max_depth; float tmin;
    uint32_t k0, k1;
    int band_rows, band_first, band_step, owned_rows;
    float* rgb_mean; int32_t* irgb;
    unsigned long long* ray_counter;
};

__device__ __forceinline__ int owned_row_to_y(const RenderArgs& a, int r)
{
    const int k = r / a.band_rows;
    return (a.band_first + k * a.band_step) * a.band_rows + (r - k * a.band_rows);
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

int pixel_render(rt_context* ctx, const RenderFrame& f)
{
    const rt_params* p = f.p;
    RenderArgs a;
    a.geom = ctx->d_geom; a.mat = ctx->d_mat; a.aux = ctx->d_aux; a.n = ctx->n; a.cam = ctx->cam;
    a.width = p->width; a.height = p->height; a.spp = p->spp; a.max_depth = p->max_depth; a.tmin = p->tmin;
    a.k0 = (uint32_t)p->seed; a.k1 = (uint32_t)(p->seed >> 32);
    a.band_rows = p->band_rows; a.band_first = p->band_first; a.band_step = p->band_step; a.owned_rows = f.owned;
    a.rgb_mean = f.d_rgb_mean; a.irgb = f.d_irgb; a.ray_counter = ctx->d_rays;
    const dim3 grid((p->width + 15) / 16, (f.owned + 15) / 16), block(kBlock);
    if (ctx->n <= kMaxLdsSpheres) hipLaunchKernelGGL(k_render_mega<true>, grid, block, (size_t)ctx->n * kLdsBytesPerSphere, f.st, a);
    else hipLaunchKernelGGL(k_render_mega<false>, grid, block, 0, f.st, a);
    HIP_TRY(ctx, hipGetLastError());
    return RT_OK;
}
