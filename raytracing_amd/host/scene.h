// RandomScene() of the reference (main.cpp:41-83) with the rand() draws of each constructor call
// sequenced explicitly in the order the pinned reference build (g++ 11.4, x86-64) evaluates them, so
// the 487-sphere scene no longer depends on the compiler: per grid cell, choose_mat, then z jitter,
// then x jitter; Lambertian albedo products drawn b, g, r (second factor first); Metal albedo b, g,
// r and then the fuzz.
#pragma once
#include <cstdlib>
#include "material.h"

inline double RtDraw() { return std::rand() / (RAND_MAX + 1.0); }

inline Hittable* RandomScene()
{
    const int n = 500;
    Hittable** list = new Hittable*[n + 1];
    int i = 0;
    list[i++] = new Sphere(Vec3(0.0f, -1000.0f, 0.0f), 1000.0f, new Lambertian(Vec3(0.5f, 0.5f, 0.5f)));
    for (int a = -11; a < 11; ++a) {
        for (int b = -11; b < 11; ++b) {
            const float chooseMat = static_cast<float>(RtDraw());
            const float z = static_cast<float>(b + 0.9f * RtDraw());
            const float x = static_cast<float>(a + 0.9f * RtDraw());
            const Vec3 center(x, 0.2f, z);
            if (!((center - Vec3(4.0f, 0.2f, 0.0f)).Length() > 0.9f)) continue;
            if (chooseMat < 0.8f) {
                const double b2 = RtDraw(), b1 = RtDraw(), g2 = RtDraw(), g1 = RtDraw(), r2 = RtDraw(), r1 = RtDraw();
                list[i++] = new Sphere(center, 0.2f, new Lambertian(Vec3(static_cast<float>(r1 * r2), static_cast<float>(g1 * g2), static_cast<float>(b1 * b2))));
            } else if (chooseMat < 0.95f) {
                const float mb = static_cast<float>(0.5f * (1.0f + RtDraw()));
                const float mg = static_cast<float>(0.5f * (1.0f + RtDraw()));
                const float mr = static_cast<float>(0.5f * (1.0f + RtDraw()));
                const float fuzz = static_cast<float>(0.5f * RtDraw());
                list[i++] = new Sphere(center, 0.2f, new Metal(Vec3(mr, mg, mb), fuzz));
            } else {
                list[i++] = new Sphere(center, 0.2f, new Dielectric(1.5f));
            }
        }
    }
    list[i++] = new Sphere(Vec3(0.0f, 1.0f, 0.0f), 1.0f, new Dielectric(1.5f));
    list[i++] = new Sphere(Vec3(-4.0f, 1.0f, 0.0f), 1.0f, new Lambertian(Vec3(0.4f, 0.2f, 0.1f)));
    list[i++] = new Sphere(Vec3(4.0f, 1.0f, 0.0f), 1.0f, new Metal(Vec3(0.7f, 0.6f, 0.5f), 0.0f));
    return new HittableList(list, i);
}

// The reference never frees its scene (main.cpp:45-82); this does.
inline void DestroyScene(Hittable* world)
{
    HittableList* l = dynamic_cast<HittableList*>(world);
    if (!l) { delete world; return; }
    for (int i = 0; i < l->m_iListSize; ++i) {
        if (Sphere* s = dynamic_cast<Sphere*>(l->m_oList[i])) delete s->m_oMaterial;
        delete l->m_oList[i];
    }
    delete[] l->m_oList;
    delete l;
}
