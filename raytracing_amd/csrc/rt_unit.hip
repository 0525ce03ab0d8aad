// Unit-level entry points: the device functions the render kernels call, run on arrays, one element per lane.
#include "rt_internal.hpp"

// ============================================================================ unit-level kernels
__global__ void k_test_sphere_hit(int n, const float* ro, const float* rd, const float* c, const float* radius,
                                  const float* tmin, const float* tmax, int32_t* hit, float* t, float* p, float* nrm)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    RayD r; r.o = mk(ro[3 * e], ro[3 * e + 1], ro[3 * e + 2]); r.d = mk(rd[3 * e], rd[3 * e + 1], rd[3 * e + 2]);
    HitD rec; rec.t = 0; rec.p = mk(0, 0, 0); rec.n = mk(0, 0, 0);
    const bool h = sphere_hit(r, mk(c[3 * e], c[3 * e + 1], c[3 * e + 2]), radius[e], tmin[e], tmax[e], rec);
    hit[e] = h ? 1 : 0; t[e] = h ? rec.t : 0.0f;
    p[3 * e] = h ? rec.p.x : 0; p[3 * e + 1] = h ? rec.p.y : 0; p[3 * e + 2] = h ? rec.p.z : 0;
    nrm[3 * e] = h ? rec.n.x : 0; nrm[3 * e + 1] = h ? rec.n.y : 0; nrm[3 * e + 2] = h ? rec.n.z : 0;
}

__global__ void k_test_scatter(int n, const int32_t* type, const float* albedo, const float* param, const float* ro,
                               const float* rd, const float* hp, const float* hn, const int32_t* draws, int per,
                               int32_t* ret, float* att, float* oo, float* od, int32_t* used)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    RayD in; in.o = mk(ro[3 * e], ro[3 * e + 1], ro[3 * e + 2]); in.d = mk(rd[3 * e], rd[3 * e + 1], rd[3 * e + 2]);
    ReplayUniforms U; U.draws = draws + (size_t)e * per; U.pos = 0;
    V3 a = mk(0, 0, 0); RayD out; out.o = mk(0, 0, 0); out.d = mk(0, 0, 0);
    const bool rv = scatter(type[e], mk(albedo[3 * e], albedo[3 * e + 1], albedo[3 * e + 2]), param[e], in,
                            mk(hp[3 * e], hp[3 * e + 1], hp[3 * e + 2]), mk(hn[3 * e], hn[3 * e + 1], hn[3 * e + 2]), U, a, out);
    ret[e] = rv ? 1 : 0; used[e] = U.pos;
    att[3 * e] = a.x; att[3 * e + 1] = a.y; att[3 * e + 2] = a.z;
    oo[3 * e] = out.o.x; oo[3 * e + 1] = out.o.y; oo[3 * e + 2] = out.o.z;
    od[3 * e] = out.d.x; od[3 * e + 1] = out.d.y; od[3 * e + 2] = out.d.z;
}

__global__ void k_test_get_ray(int n, const CameraD cam, const float* s, const float* t, const float* disk, float* ro, float* rd)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const RayD r = get_ray(cam, s[e], t[e], disk[2 * e], disk[2 * e + 1]);
    ro[3 * e] = r.o.x; ro[3 * e + 1] = r.o.y; ro[3 * e + 2] = r.o.z;
    rd[3 * e] = r.d.x; rd[3 * e + 1] = r.d.y; rd[3 * e + 2] = r.d.z;
}

__global__ void k_test_helpers(int n, const float* v, const float* nrm, const float* nint, const float* cosine, const float* ref,
                               float* refl, int32_t* ok, float* refr, float* sch)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const V3 vin = mk(v[3 * e], v[3 * e + 1], v[3 * e + 2]), nn = mk(nrm[3 * e], nrm[3 * e + 1], nrm[3 * e + 2]);
    const V3 r = reflect(vin, nn);
    refl[3 * e] = r.x; refl[3 * e + 1] = r.y; refl[3 * e + 2] = r.z;
    V3 t = mk(0, 0, 0);
    const bool good = refract(vin, nn, nint[e], t);
    ok[e] = good ? 1 : 0;
    refr[3 * e] = good ? t.x : 0; refr[3 * e + 1] = good ? t.y : 0; refr[3 * e + 2] = good ? t.z : 0;
    sch[e] = schlick(cosine[e], ref[e]);
}

__global__ void k_test_philox(int n, const uint32_t* ctr, uint32_t k0, uint32_t k1, uint32_t* out)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const U4 w = philox4x32_10(ctr[4 * e], ctr[4 * e + 1], ctr[4 * e + 2], ctr[4 * e + 3], k0, k1);
    out[4 * e] = w.x; out[4 * e + 1] = w.y; out[4 * e + 2] = w.z; out[4 * e + 3] = w.w;
}

extern "C" {

// ---------------------------------------------------------------------------- unit-level entry points

int rt_test_sphere_hit(rt_context* ctx, int n, const float* ray_o, const float* ray_d, const float* center, const float* radius,
                       const float* tmin, const float* tmax, int32_t* hit, float* t, float* p, float* normal)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "rt_test_sphere_hit: null context");
    if (n <= 0 || !ray_o || !ray_d || !center || !radius || !tmin || !tmax || !hit || !t || !p || !normal) return fail(ctx, RT_ERR_INVALID, "rt_test_sphere_hit: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf o, d, c, r, a, b, oh, ot, op, on;
    const size_t f3 = sizeof(float) * 3 * n, f1 = sizeof(float) * n;
    HIP_TRY(ctx, o.up(ray_o, f3)); HIP_TRY(ctx, d.up(ray_d, f3)); HIP_TRY(ctx, c.up(center, f3)); HIP_TRY(ctx, r.up(radius, f1));
    HIP_TRY(ctx, a.up(tmin, f1)); HIP_TRY(ctx, b.up(tmax, f1));
    HIP_TRY(ctx, oh.up(nullptr, sizeof(int32_t) * n)); HIP_TRY(ctx, ot.up(nullptr, f1)); HIP_TRY(ctx, op.up(nullptr, f3)); HIP_TRY(ctx, on.up(nullptr, f3));
    hipLaunchKernelGGL(k_test_sphere_hit, dim3((n + 63) / 64), dim3(64), 0, 0, n, o.as<float>(), d.as<float>(), c.as<float>(), r.as<float>(),
                       a.as<float>(), b.as<float>(), oh.as<int32_t>(), ot.as<float>(), op.as<float>(), on.as<float>());
    HIP_TRY(ctx, hipGetLastError()); HIP_TRY(ctx, hipDeviceSynchronize());
    HIP_TRY(ctx, oh.down(hit, sizeof(int32_t) * n)); HIP_TRY(ctx, ot.down(t, f1)); HIP_TRY(ctx, op.down(p, f3)); HIP_TRY(ctx, on.down(normal, f3));
    return RT_OK;
}

int rt_test_scatter(rt_context* ctx, int n, const int32_t* material, const float* albedo_rgb, const float* param, const float* ray_o,
                    const float* ray_d, const float* hit_p, const float* hit_n, const int32_t* draws, int draws_per_case,
                    int32_t* ret, float* attenuation, float* out_o, float* out_d, int32_t* used)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "rt_test_scatter: null context");
    if (n <= 0 || draws_per_case <= 0 || !material || !albedo_rgb || !param || !ray_o || !ray_d || !hit_p || !hit_n || !draws || !ret || !attenuation || !out_o || !out_d || !used)
        return fail(ctx, RT_ERR_INVALID, "rt_test_scatter: bad argument");
    for (int i = 0; i < n; ++i) if (material[i] < 0 || material[i] > 2) return fail(ctx, RT_ERR_INVALID, "rt_test_scatter: unknown material %d", material[i]);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf m, al, pa, o, d, hp, hn, dr, r, at, oo, od, us;
    const size_t f3 = sizeof(float) * 3 * n, f1 = sizeof(float) * n, i1 = sizeof(int32_t) * n;
    HIP_TRY(ctx, m.up(material, i1)); HIP_TRY(ctx, al.up(albedo_rgb, f3)); HIP_TRY(ctx, pa.up(param, f1)); HIP_TRY(ctx, o.up(ray_o, f3));
    HIP_TRY(ctx, d.up(ray_d, f3)); HIP_TRY(ctx, hp.up(hit_p, f3)); HIP_TRY(ctx, hn.up(hit_n, f3)); HIP_TRY(ctx, dr.up(draws, sizeof(int32_t) * (size_t)n * draws_per_case));
    HIP_TRY(ctx, r.up(nullptr, i1)); HIP_TRY(ctx, at.up(nullptr, f3)); HIP_TRY(ctx, oo.up(nullptr, f3)); HIP_TRY(ctx, od.up(nullptr, f3)); HIP_TRY(ctx, us.up(nullptr, i1));
    hipLaunchKernelGGL(k_test_scatter, dim3((n + 63) / 64), dim3(64), 0, 0, n, m.as<int32_t>(), al.as<float>(), pa.as<float>(), o.as<float>(), d.as<float>(),
                       hp.as<float>(), hn.as<float>(), dr.as<int32_t>(), draws_per_case, r.as<int32_t>(), at.as<float>(), oo.as<float>(), od.as<float>(), us.as<int32_t>());
    HIP_TRY(ctx, hipGetLastError()); HIP_TRY(ctx, hipDeviceSynchronize());
    HIP_TRY(ctx, r.down(ret, i1)); HIP_TRY(ctx, at.down(attenuation, f3)); HIP_TRY(ctx, oo.down(out_o, f3)); HIP_TRY(ctx, od.down(out_d, f3)); HIP_TRY(ctx, us.down(used, i1));
    return RT_OK;
}

int rt_test_get_ray(rt_context* ctx, int n, const rt_camera_pod* cam, const float* s, const float* t, const float* disk_xy, float* ray_o, float* ray_d)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "rt_test_get_ray: null context");
    if (n <= 0 || !cam || !s || !t || !disk_xy || !ray_o || !ray_d) return fail(ctx, RT_ERR_INVALID, "rt_test_get_ray: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf ds, dt, dd, o, d;
    const size_t f3 = sizeof(float) * 3 * n, f1 = sizeof(float) * n;
    HIP_TRY(ctx, ds.up(s, f1)); HIP_TRY(ctx, dt.up(t, f1)); HIP_TRY(ctx, dd.up(disk_xy, 2 * f1)); HIP_TRY(ctx, o.up(nullptr, f3)); HIP_TRY(ctx, d.up(nullptr, f3));
    hipLaunchKernelGGL(k_test_get_ray, dim3((n + 63) / 64), dim3(64), 0, 0, n, to_device(*cam), ds.as<float>(), dt.as<float>(), dd.as<float>(), o.as<float>(), d.as<float>());
    HIP_TRY(ctx, hipGetLastError()); HIP_TRY(ctx, hipDeviceSynchronize());
    HIP_TRY(ctx, o.down(ray_o, f3)); HIP_TRY(ctx, d.down(ray_d, f3));
    return RT_OK;
}

int rt_test_helpers(rt_context* ctx, int n, const float* v, const float* normal, const float* ni_over_nt, const float* cosine,
                    const float* ref_idx, float* reflected, int32_t* refract_ok, float* refracted, float* schlick_out)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "rt_test_helpers: null context");
    if (n <= 0 || !v || !normal || !ni_over_nt || !cosine || !ref_idx || !reflected || !refract_ok || !refracted || !schlick_out) return fail(ctx, RT_ERR_INVALID, "rt_test_helpers: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf dv, dn, di, dc, dr, orf, ook, ort, os;
    const size_t f3 = sizeof(float) * 3 * n, f1 = sizeof(float) * n;
    HIP_TRY(ctx, dv.up(v, f3)); HIP_TRY(ctx, dn.up(normal, f3)); HIP_TRY(ctx, di.up(ni_over_nt, f1)); HIP_TRY(ctx, dc.up(cosine, f1)); HIP_TRY(ctx, dr.up(ref_idx, f1));
    HIP_TRY(ctx, orf.up(nullptr, f3)); HIP_TRY(ctx, ook.up(nullptr, sizeof(int32_t) * n)); HIP_TRY(ctx, ort.up(nullptr, f3)); HIP_TRY(ctx, os.up(nullptr, f1));
    hipLaunchKernelGGL(k_test_helpers, dim3((n + 63) / 64), dim3(64), 0, 0, n, dv.as<float>(), dn.as<float>(), di.as<float>(), dc.as<float>(), dr.as<float>(),
                       orf.as<float>(), ook.as<int32_t>(), ort.as<float>(), os.as<float>());
    HIP_TRY(ctx, hipGetLastError()); HIP_TRY(ctx, hipDeviceSynchronize());
    HIP_TRY(ctx, orf.down(reflected, f3)); HIP_TRY(ctx, ook.down(refract_ok, sizeof(int32_t) * n)); HIP_TRY(ctx, ort.down(refracted, f3)); HIP_TRY(ctx, os.down(schlick_out, f1));
    return RT_OK;
}

int rt_test_philox(rt_context* ctx, int n, const uint32_t* ctr, const uint32_t* key, uint32_t* out)
{
    if (!ctx) return fail(nullptr, RT_ERR_INVALID, "rt_test_philox: null context");
    if (n <= 0 || !ctr || !key || !out) return fail(ctx, RT_ERR_INVALID, "rt_test_philox: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf c, o;
    HIP_TRY(ctx, c.up(ctr, sizeof(uint32_t) * 4 * n)); HIP_TRY(ctx, o.up(nullptr, sizeof(uint32_t) * 4 * n));
    hipLaunchKernelGGL(k_test_philox, dim3((n + 63) / 64), dim3(64), 0, 0, n, c.as<uint32_t>(), key[0], key[1], o.as<uint32_t>());
    HIP_TRY(ctx, hipGetLastError()); HIP_TRY(ctx, hipDeviceSynchronize());
    HIP_TRY(ctx, o.down(out, sizeof(uint32_t) * 4 * n));
    return RT_OK;
}

}  // extern "C"
