// Host mirror of the reference's materials (material.h:7-186): Schlick / Refract / Reflect /
// RandomInUnitSphere and Lambertian / Metal / Dielectric with their public members.  Unlike the
// reference header, every definition is inline, so this header can be included from any number of
// translation units.  Scatter() keeps the reference's CPU arithmetic and rand() usage; the constructor
// argument draws are sequenced explicitly in the order the pinned reference build evaluates them.
#pragma once
#include <cstdlib>
#include "hittable.h"

inline float Schlick(float cosine, float ref_idx)
{
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    return static_cast<float>(r0 + (1.0f - r0) * std::pow(static_cast<double>(1.0f - cosine), 5.0));
}

inline bool Refract(const Vec3& v, const Vec3& n, float ni_over_nt, Vec3& refracted)
{
    const Vec3 uv = Unit_Vector(v);
    const float dt = Dot(uv, n);
    const float disc = 1.0f - ni_over_nt * ni_over_nt * (1.0f - dt * dt);
    if (disc > 0) { refracted = ni_over_nt * (uv - n * dt) - n * sqrtf(disc); return true; }
    return false;
}

inline Vec3 Reflect(const Vec3& v, const Vec3& n) { return v - 2 * Dot(v, n) * n; }

inline float RtUniform() { return static_cast<float>(std::rand() / (RAND_MAX + 1.0)); }

inline Vec3 RandomInUnitSphere()
{
    Vec3 p;
    do {
        const float z = RtUniform(), y = RtUniform(), x = RtUniform();   // right-to-left, as the pinned build draws
        p = 2.0f * Vec3(x, y, z) - Vec3(1.0f, 1.0f, 1.0f);
    } while (p.SquaredLength() >= 1.0f);
    return p;
}

class Material {
public:
    virtual ~Material() = default;
    virtual bool Scatter(const Ray& in, const HitRecord& rec, Vec3& attenuation, Ray& scattered) const = 0;
};

class Lambertian : public Material {
public:
    Vec3 m_oAlbedo;
    explicit Lambertian(const Vec3& a) : m_oAlbedo(a) {}
    bool Scatter(const Ray&, const HitRecord& rec, Vec3& attenuation, Ray& scattered) const override
    {
        const Vec3 target = rec.m_oPoint + rec.m_oNormal + RandomInUnitSphere();
        scattered = Ray(rec.m_oPoint, target - rec.m_oPoint);
        attenuation = m_oAlbedo;
        return true;
    }
};

class Metal : public Material {
public:
    Vec3 m_oAlbedo;
    float m_fFuzz;
    Metal(const Vec3& a, float fuzz) : m_oAlbedo(a), m_fFuzz(fuzz < 1 ? fuzz : 1.0f) {}
    bool Scatter(const Ray& in, const HitRecord& rec, Vec3& attenuation, Ray& scattered) const override
    {
        const Vec3 reflected = Reflect(Unit_Vector(in.Direction()), rec.m_oNormal);
        scattered = Ray(rec.m_oPoint, reflected + m_fFuzz * RandomInUnitSphere());
        attenuation = m_oAlbedo;
        return Dot(scattered.Direction(), rec.m_oNormal) > 0;
    }
};

class Dielectric : public Material {
public:
    float m_fRefIdx;
    explicit Dielectric(float ri) : m_fRefIdx(ri) {}
    bool Scatter(const Ray& in, const HitRecord& rec, Vec3& attenuation, Ray& scattered) const override
    {
        const Vec3 d = in.Direction(), n = rec.m_oNormal;
        const Vec3 reflected = Reflect(d, n);
        attenuation = Vec3(1.0f, 1.0f, 1.0f);
        const float dn = Dot(d, n);
        const bool leaving = dn > 0;
        const Vec3 outward = leaving ? -n : n;
        const float ni_over_nt = leaving ? m_fRefIdx : 1.0f / m_fRefIdx;
        float cosine;
        if (leaving) {
            cosine = dn / d.Length();
            cosine = sqrtf(1.0f - m_fRefIdx * m_fRefIdx * (1.0f - cosine * cosine));
        } else {
            cosine = -dn / d.Length();
        }
        Vec3 refracted(0.0f, 0.0f, 0.0f);
        const float reflect_prob = Refract(d, outward, ni_over_nt, refracted) ? Schlick(cosine, m_fRefIdx) : 1.0f;
        scattered = Ray(rec.m_oPoint, (std::rand() / (RAND_MAX + 1.0)) < reflect_prob ? reflected : refracted);
        return true;
    }
};
