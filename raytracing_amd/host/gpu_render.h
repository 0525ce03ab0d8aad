// The seam between the reference-shaped host code and the GPU library: flatten a HittableList of
// Spheres into plain arrays, copy the Camera's fields, make one render call (rt_amd.h), get back the
// integers main.cpp:138-140 would have produced.  Errors are reported by exception here (host C++);
// the C ABI underneath reports them by status code.
#pragma once
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "camera.h"
#include "material.h"
#include "rt_amd.h"

namespace rtamd {

struct FlatScene {
    std::vector<float> cx, cy, cz, radius, albedo, param;
    std::vector<int32_t> material;
    int size() const { return (int)cx.size(); }
};

// Walks HittableList::m_oList.  The reference's Material has no type tag, so the concrete class is
// recovered with dynamic_cast; anything that is not a Sphere with one of the three materials is an error.
inline FlatScene Flatten(const Hittable* world)
{
    const HittableList* list = dynamic_cast<const HittableList*>(world);
    if (!list) throw std::invalid_argument("Flatten: world is not a HittableList");
    FlatScene f;
    for (int i = 0; i < list->m_iListSize; ++i) {
        const Sphere* s = dynamic_cast<const Sphere*>(list->m_oList[i]);
        if (!s) throw std::invalid_argument("Flatten: element " + std::to_string(i) + " is not a Sphere");
        f.cx.push_back(s->m_oCenter[0]); f.cy.push_back(s->m_oCenter[1]); f.cz.push_back(s->m_oCenter[2]);
        f.radius.push_back(s->m_fRadius);
        Vec3 albedo(1.0f, 1.0f, 1.0f); float param = 0.0f; int32_t type;
        if (const Lambertian* l = dynamic_cast<const Lambertian*>(s->m_oMaterial)) { type = RT_LAMBERTIAN; albedo = l->m_oAlbedo; }
        else if (const Metal* m = dynamic_cast<const Metal*>(s->m_oMaterial)) { type = RT_METAL; albedo = m->m_oAlbedo; param = m->m_fFuzz; }
        else if (const Dielectric* d = dynamic_cast<const Dielectric*>(s->m_oMaterial)) { type = RT_DIELECTRIC; param = d->m_fRefIdx; }
        else throw std::invalid_argument("Flatten: sphere " + std::to_string(i) + " has an unknown material");
        f.material.push_back(type);
        for (int k = 0; k < 3; ++k) f.albedo.push_back(albedo[k]);
        f.param.push_back(param);
    }
    return f;
}

inline rt_camera_pod ToPod(const Camera& c)
{
    rt_camera_pod p;
    for (int k = 0; k < 3; ++k) {
        p.origin[k] = c.m_oOrigin[k]; p.lower_left_corner[k] = c.m_oLowerLeftCorner[k];
        p.horizontal[k] = c.m_oHorizontal[k]; p.vertical[k] = c.m_oVertical[k];
        p.u[k] = c.m_oU[k]; p.v[k] = c.m_oV[k]; p.w[k] = c.m_oW[k];
    }
    p.lens_radius = c.m_fLensRadius;
    return p;
}

struct RenderResult { std::vector<int32_t> irgb; uint64_t rays = 0; float kernel_ms = 0.0f; };

// A GPU context holding one flattened scene and one camera; renders whole images or row bands of them.
class GpuRenderer {
public:
    GpuRenderer(const Hittable* world, const Camera& cam, int device = 0)
    {
        if (rt_create(device, &m_ctx) != RT_OK) throw std::runtime_error(std::string("rt_create: ") + rt_last_error(nullptr));
        try {
            const FlatScene f = Flatten(world);
            Check(rt_upload_scene(m_ctx, f.size(), f.cx.data(), f.cy.data(), f.cz.data(), f.radius.data(), f.material.data(), f.albedo.data(), f.param.data()), "rt_upload_scene");
            const rt_camera_pod pod = ToPod(cam);
            Check(rt_set_camera(m_ctx, &pod), "rt_set_camera");
        } catch (...) { rt_destroy(m_ctx); throw; }
    }
    ~GpuRenderer() { rt_destroy(m_ctx); }
    GpuRenderer(const GpuRenderer&) = delete;
    GpuRenderer& operator=(const GpuRenderer&) = delete;

    static rt_params Params(int nx, int ny, int ns, uint64_t seed)
    {
        rt_params p{};
        p.width = nx; p.height = ny; p.spp = ns; p.max_depth = 50; p.tmin = 0.001f; p.seed = seed;
        p.band_rows = ny; p.band_first = 0; p.band_step = 1; p.mode = RT_MODE_MEGAKERNEL;
        return p;
    }

    // The bands p selects; irgb must hold 3 * rt_owned_pixels(&p) ints.  Returns the kernel time in ms and
    // adds the call's counters to *counters when given.
    float Render(const rt_params& p, int32_t* irgb, rt_counters* counters = nullptr)
    {
        float ms = 0.0f;
        Check(rt_render(m_ctx, &p, nullptr, irgb, nullptr, &ms), "rt_render");
        if (counters) {
            rt_counters c{};
            Check(rt_render_counters(m_ctx, &c), "rt_render_counters");
            counters->rays_traced += c.rays_traced;
            counters->intersection_tests += c.intersection_tests;
            counters->material_evaluations += c.material_evaluations;
        }
        return ms;
    }

    // Progressive form: the image is rendered in bands of band_rows rows, top to bottom; after each band
    // on_band(first_row, rows, irgb_of_the_band) is called, so a caller can show or store finished rows
    // while the rest is still being rendered.  Because every uniform is keyed on the pixel, the rows are
    // the same as those of a single whole-image call.
    template <class OnBand>
    float RenderProgressive(int nx, int ny, int ns, uint64_t seed, int band_rows, rt_counters* counters, OnBand on_band)
    {
        if (band_rows <= 0) throw std::invalid_argument("RenderProgressive: band_rows must be positive");
        const int bands = (ny + band_rows - 1) / band_rows;
        std::vector<int32_t> band((size_t)band_rows * nx * 3);
        float total_ms = 0.0f;
        for (int b = 0; b < bands; ++b) {
            rt_params p = Params(nx, ny, ns, seed);
            p.band_rows = band_rows; p.band_first = b; p.band_step = bands;     // exactly band b
            total_ms += Render(p, band.data(), counters);
            on_band(b * band_rows, (int)rt_owned_rows(&p), band.data());
        }
        return total_ms;
    }

    rt_context* Context() const { return m_ctx; }

private:
    void Check(int rc, const char* what) const
    {
        if (rc != RT_OK) throw std::runtime_error(std::string(what) + ": " + rt_last_error(m_ctx));
    }
    rt_context* m_ctx = nullptr;
};

// Replaces the nested loop of main.cpp:118-141 for the whole image on GPU `device`.
inline RenderResult RenderOnGpu(const Hittable* world, const Camera& cam, int nx, int ny, int ns, uint64_t seed = 0, int device = 0)
{
    GpuRenderer gpu(world, cam, device);
    RenderResult out;
    out.irgb.resize((size_t)nx * ny * 3);
    rt_counters c{};
    out.kernel_ms = gpu.Render(GpuRenderer::Params(nx, ny, ns, seed), out.irgb.data(), &c);
    out.rays = c.rays_traced;
    return out;
}

// The same image over `devices` GPUs of the node (0 = all that are visible): one context per GPU, every one holding the
// scene and the camera, single rows dealt round-robin (rt_render_multi).  The thread pool over tiles of
// main_modern.cpp:197-219 with GPUs as the workers.
inline RenderResult RenderOnGpus(const Hittable* world, const Camera& cam, int nx, int ny, int ns, uint64_t seed, int devices)
{
    int visible = 0;
    if (rt_device_count(&visible) != RT_OK || visible <= 0) throw std::runtime_error(std::string("rt_device_count: ") + rt_last_error(nullptr));
    if (devices <= 0) devices = visible;
    if (devices > visible) throw std::invalid_argument("RenderOnGpus: " + std::to_string(devices) + " GPUs asked for, " + std::to_string(visible) + " visible");
    std::vector<std::unique_ptr<GpuRenderer>> gpus;
    for (int d = 0; d < devices; ++d) gpus.emplace_back(new GpuRenderer(world, cam, d));
    std::vector<rt_context*> ctx;
    for (auto& g : gpus) ctx.push_back(g->Context());
    RenderResult out;
    out.irgb.resize((size_t)nx * ny * 3);
    rt_params p = GpuRenderer::Params(nx, ny, ns, seed);
    p.band_rows = 1;
    std::vector<float> ms(devices, 0.0f);
    if (rt_render_multi(ctx.data(), devices, &p, nullptr, out.irgb.data(), &out.rays, ms.data()) != RT_OK)
        throw std::runtime_error(std::string("rt_render_multi: ") + rt_last_error(ctx[0]));
    for (float m : ms) out.kernel_ms = m > out.kernel_ms ? m : out.kernel_ms;       // the slowest GPU
    return out;
}

}  // namespace rtamd
