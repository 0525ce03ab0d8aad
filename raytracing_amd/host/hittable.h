// Host mirror of HitRecord / Hittable (hittable.h:8-21), Sphere (sphere.h:6-18, sphere.cpp:15-52) and
// HittableList (hittablelist.h:6-16, hittablelist.cpp:13-31).  The Hit() bodies are the reference's
// CPU arithmetic, kept so that host code written against the reference keeps working; the render
// loop itself never calls them (it runs on the GPU through rt_amd.h).
#pragma once
#include "ray.h"

class Material;

struct HitRecord {
    float m_fT;
    Vec3 m_oPoint;
    Vec3 m_oNormal;
    Material* m_oMaterial;
};

class Hittable {
public:
    virtual ~Hittable() = default;
    virtual bool Hit(const Ray& r, float tmin, float tmax, HitRecord& rec) const = 0;
};

class Sphere : public Hittable {
public:
    Vec3 m_oCenter;
    float m_fRadius;
    Material* m_oMaterial;

    Sphere() : m_oCenter(Vec3()), m_fRadius(0.0f), m_oMaterial(nullptr) {}
    Sphere(Vec3 c, float radius, Material* m) : m_oCenter(c), m_fRadius(radius), m_oMaterial(m) {}

    bool Hit(const Ray& r, float tmin, float tmax, HitRecord& rec) const override
    {
        const Vec3 oc = r.Origin() - m_oCenter;
        const float a = Dot(r.Direction(), r.Direction());
        const float b = Dot(oc, r.Direction());
        const float c = Dot(oc, oc) - m_fRadius * m_fRadius;
        const float desc = b * b - a * c;
        if (!(desc > 0.0f)) return false;
        const float roots[2] = { (-b - sqrtf(desc)) / a, (-b + sqrtf(desc)) / a };
        for (float t : roots) {
            if (t < tmax && t > tmin) {
                rec.m_fT = t;
                rec.m_oPoint = r.PointAtParamenter(t);
                rec.m_oNormal = (rec.m_oPoint - m_oCenter) / m_fRadius;
                rec.m_oMaterial = m_oMaterial;
                return true;
            }
        }
        return false;
    }
};

class HittableList : public Hittable {
public:
    Hittable** m_oList;
    int m_iListSize;

    HittableList() : m_oList(nullptr), m_iListSize(0) {}
    HittableList(Hittable** l, int n) : m_oList(l), m_iListSize(n) {}

    bool Hit(const Ray& r, float tmin, float tmax, HitRecord& rec) const override
    {
        HitRecord tmp;
        bool any = false;
        float closest = tmax;
        for (int i = 0; i < m_iListSize; ++i) {
            if (m_oList[i]->Hit(r, tmin, closest, tmp)) { any = true; closest = tmp.m_fT; rec = tmp; }
        }
        return any;
    }
};
