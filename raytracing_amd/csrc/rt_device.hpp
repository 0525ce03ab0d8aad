// Device-side arithmetic of the render loop for gfx950.
//
// Exactness contract: built with -ffp-contract=off, every function here performs the same IEEE
// binary32 operations, in the same order, as the reference's C++ (vec3.h:69-135, sphere.cpp:15-52,
// material.h:7-186, camera.cpp:46-53, main.cpp:13-39).  hipcc's default correctly-rounded sqrt and
// divide are relied on.  Wherever a fused multiply-add is wanted it is written as fmaf() in a
// code path whose result is only ever used as a conservative filter.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt {

struct V3 { float x, y, z; };
struct RayD { V3 o, d; };

__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 v; v.x = x; v.y = y; v.z = z; return v; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator*(float k, V3 a) { return mk(k * a.x, k * a.y, k * a.z); }
__device__ __forceinline__ V3 operator*(V3 a, float k) { return mk(a.x * k, a.y * k, a.z * k); }
__device__ __forceinline__ V3 operator/(V3 a, float k) { return mk(a.x / k, a.y / k, a.z / k); }
__device__ __forceinline__ V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
__device__ __forceinline__ float sqlen(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
__device__ __forceinline__ float length(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
__device__ __forceinline__ V3 unit(V3 a) { return a / length(a); }

// ---------------------------------------------------------------------------- Philox4x32-10
// Counter (pixel, sample, bounce, group), key = seed.  One call yields the four words of a group.
struct U4 { uint32_t x, y, z, w; };
// kWide: both halves of a product from one 64-bit multiply (v_mad_u64_u32) instead of a high and a low multiply -- half the
// multiplier work, at the price of 64-bit register pairs; for code with registers to spare.
template <bool kWide>
__device__ __forceinline__ U4 philox4x32_10_t(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t h0, l0, h1, l1;
        if (kWide) {
            const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
            h0 = (uint32_t)(p0 >> 32); l0 = (uint32_t)p0; h1 = (uint32_t)(p1 >> 32); l1 = (uint32_t)p1;
        } else {
            h0 = __umulhi(0xD2511F53u, c0); l0 = 0xD2511F53u * c0;
            h1 = __umulhi(0xCD9E8D57u, c2); l1 = 0xCD9E8D57u * c2;
        }
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    U4 o; o.x = c0; o.y = c1; o.z = c2; o.w = c3; return o;
}
__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) { return philox4x32_10_t<false>(c0, c1, c2, c3, k0, k1); }
// 24-bit uniform in [0,1): exactly representable, equal to rand()/(RAND_MAX+1.0) for rand() = (w>>8)<<7.
__device__ __forceinline__ float u01(uint32_t w) { return (float)(w >> 8) * (1.0f / 16777216.0f); }

// Uniform providers.  f() is `static_cast<float>(rand()/(RAND_MAX+1.0))`, d() the double itself.
// Keyed provider: draw d of the current slot is word d%4 of group d/4 (all four words of a Philox
// block are consumed before the next block is computed).
template <bool kWide>
struct KeyedUniformsT {
    uint32_t px, smp, bounce, k0, k1; U4 cur;
    __device__ __forceinline__ void slot(uint32_t p, uint32_t s, uint32_t b) { px = p; smp = s; bounce = b; }
    // Computes group 0 of the current slot (shared by whichever material consumes it).
    __device__ __forceinline__ void prefetch() { cur = philox4x32_10_t<kWide>(px, smp, bounce, 0u, k0, k1); }
    // First draw of the slot (Dielectric::Scatter's only draw).
    __device__ __forceinline__ float f() { return u01(cur.x); }
    __device__ __forceinline__ double d() { return (double)f(); }
};
using KeyedUniforms = KeyedUniformsT<false>;
// Replay provider for the unit-level entry points: recorded rand() integers.
struct ReplayUniforms {
    const int32_t* draws; int pos;
    __device__ __forceinline__ void prefetch() {}
    __device__ __forceinline__ double d() { return (double)draws[pos++] / 2147483648.0; }
    __device__ __forceinline__ float f() { return (float)d(); }
};

// ---------------------------------------------------------------------------- Sphere::Hit
// Literal form (sphere.cpp:15-52): returns the accepted root or a negative value.
struct HitD { float t; V3 p, n; };
__device__ __forceinline__ bool sphere_hit(const RayD& r, V3 c, float radius, float tmin, float tmax, HitD& rec)
{
    const V3 oc = r.o - c;
    const float a = dot(r.d, r.d);
    const float b = dot(oc, r.d);
    const float cc = dot(oc, oc) - radius * radius;
    const float desc = b * b - a * cc;
    if (desc > 0.0f) {
        const float sq = sqrtf(desc);
        float temp = (-b - sq) / a;
        if (!(temp < tmax && temp > tmin)) temp = (-b + sq) / a;
        if (temp < tmax && temp > tmin) {
            rec.t = temp;
            rec.p = r.o + temp * r.d;
            rec.n = (rec.p - c) / radius;
            return true;
        }
    }
    return false;
}

// ---------------------------------------------------------------------------- material.h helpers
__device__ __forceinline__ V3 reflect(V3 v, V3 n) { return v - (2 * dot(v, n)) * n; }
__device__ __forceinline__ bool refract(V3 v, V3 n, float ni_over_nt, V3& out)
{
    const V3 uv = unit(v);
    const float dt = dot(uv, n);
    const float disc = 1.0f - ni_over_nt * ni_over_nt * (1.0f - dt * dt);
    if (disc > 0) { out = ni_over_nt * (uv - n * dt) - n * sqrtf(disc); return true; }
    return false;
}
__device__ __forceinline__ float schlick(float cosine, float ref_idx)
{
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    // pow(1 - cosine, 5) is evaluated in double by the reference; x^5 by three double products
    // differs from libm's pow by at most a few double ulps, far below float resolution.
    const double x = (double)(1.0f - cosine), x2 = x * x;
    return (float)((double)r0 + (double)(1.0f - r0) * (x2 * x2 * x));
}
// Keyed stream: the draws of RandomInUnitSphere() are words 0,1,2 | 3,4,5 | ... of the slot, i.e. four
// rejection iterations per three Philox blocks.  Written out with static word selection (a runtime
// word index would put the block into scratch memory).  u.cur must hold group 0 (prefetch()).
__device__ __forceinline__ bool unit_ball_try(uint32_t wz, uint32_t wy, uint32_t wx, V3& p)
{
    p = 2.0f * mk(u01(wx), u01(wy), u01(wz)) - mk(1.0f, 1.0f, 1.0f);   // first draw is z: right-to-left argument order
    return !(sqlen(p) >= 1.0f);
}
template <bool kWide>
__device__ __forceinline__ V3 random_in_unit_sphere(KeyedUniformsT<kWide>& u)
{
    V3 p;
    U4 b0 = u.cur;
    for (uint32_t g = 0;; g += 3) {
        if (g != 0) b0 = philox4x32_10_t<kWide>(u.px, u.smp, u.bounce, g, u.k0, u.k1);
        if (unit_ball_try(b0.x, b0.y, b0.z, p)) break;
        const U4 b1 = philox4x32_10_t<kWide>(u.px, u.smp, u.bounce, g + 1u, u.k0, u.k1);
        if (unit_ball_try(b0.w, b1.x, b1.y, p)) break;
        const U4 b2 = philox4x32_10_t<kWide>(u.px, u.smp, u.bounce, g + 2u, u.k0, u.k1);
        if (unit_ball_try(b1.z, b1.w, b2.x, p)) break;
        if (unit_ball_try(b2.y, b2.z, b2.w, p)) break;
    }
    return p;
}
template <class U>
__device__ __forceinline__ V3 random_in_unit_sphere(U& u)
{
    V3 p;
    do {
        // the pinned reference build evaluates the three constructor arguments right to left
        const float z = u.f(), y = u.f(), x = u.f();
        p = 2.0f * mk(x, y, z) - mk(1.0f, 1.0f, 1.0f);
    } while (sqlen(p) >= 1.0f);
    return p;
}

// Material::Scatter for the three materials.  `param` = fuzz | ref_idx.  Lambertian and Metal both
// draw one RandomInUnitSphere() and nothing else, so the rejection loop is shared by the lanes of
// either kind; the arithmetic that follows is the reference's, material by material.
template <class U>
__device__ __forceinline__ bool scatter(int type, V3 albedo, float param, const RayD& in, V3 p, V3 n, U& u, V3& att, RayD& out)
{
    out.o = p;
    u.prefetch();
    if (type != 2) {
        const V3 rnd = random_in_unit_sphere(u);
        att = albedo;
        if (type == 0) {
            const V3 target = (p + n) + rnd;
            out.d = target - p;
            return true;
        }
        const V3 reflected = reflect(unit(in.d), n);
        out.d = reflected + param * rnd;
        return dot(out.d, n) > 0;
    }
    const V3 reflected = reflect(in.d, n);
    att = mk(1.0f, 1.0f, 1.0f);
    V3 outward, refracted = mk(0.0f, 0.0f, 0.0f);
    float ni_over_nt, cosine, reflect_prob;
    const float dn = dot(in.d, n);
    if (dn > 0) {
        outward = -n;
        ni_over_nt = param;
        cosine = dn / length(in.d);
        cosine = sqrtf(1.0f - param * param * (1.0f - cosine * cosine));
    } else {
        outward = n;
        ni_over_nt = 1.0f / param;
        cosine = -dn / length(in.d);
    }
    if (refract(in.d, outward, ni_over_nt, refracted)) reflect_prob = schlick(cosine, param);
    else reflect_prob = 1.0f;
    out.d = (u.d() < (double)reflect_prob) ? reflected : refracted;
    return true;
}

// Lens sample t = 0, 1, 2, ... of slot (pixel, sample, 0) (RandomUnitInDisk's rejection loop,
// camera.cpp:31-44): try 0 is words 2,3 of group 0 (whose words 0,1 are the pixel jitter); try
// t >= 1 is words 0,1 (t odd) or 2,3 (t even) of group (t + 1) / 2.
__device__ __forceinline__ void lens_words(const U4& blk, uint32_t t, float& a, float& b)
{
    const bool hi = (t == 0) || ((t & 1u) == 0);
    a = u01(hi ? blk.z : blk.x); b = u01(hi ? blk.w : blk.y);
}

// ---------------------------------------------------------------------------- Camera::GetRay
struct CameraD { V3 origin, llc, horiz, vert, u, v; float lens_radius; };
__device__ __forceinline__ RayD get_ray(const CameraD& c, float s, float t, float disk_x, float disk_y)
{
    const V3 rd = c.lens_radius * mk(disk_x, disk_y, 0.0f);
    const V3 offset = c.u * rd.x + c.v * rd.y;
    RayD r;
    r.o = c.origin + offset;
    r.d = (((c.llc + s * c.horiz) + t * c.vert) - c.origin) - offset;
    return r;
}

// Sky gradient of Color()'s miss branch (main.cpp:33-37).
__device__ __forceinline__ V3 sky(V3 dir)
{
    const V3 ud = unit(dir);
    const float t = 0.5f * (ud.y + 1.0f);
    return (1.0f - t) * mk(1.0f, 1.0f, 1.0f) + t * mk(0.5f, 0.7f, 1.0f);
}

// int(255.99 * c) on a double product (main.cpp:138-140); NaN maps to INT_MIN as x86 does.
__device__ __forceinline__ int quantise(float mean)
{
    const double v = 255.99 * (double)sqrtf(mean);
    return (v != v) ? (int)0x80000000 : (int)v;
}

}  // namespace rt
