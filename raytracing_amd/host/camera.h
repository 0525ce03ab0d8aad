// Thin-lens camera under the reference's name and field names (camera.h:6-22), so that host code written against
// the reference compiles unchanged and rtamd::ToPod() can copy the eight fields one to one.  The constructor performs
// the float operations of camera.cpp:6-29 in the same order (the fields must come out bit-identical: the GPU render
// reads them); it is laid out as "frame, then image plane" rather than as the reference's single run of statements.
// GetRay() is the host-side form of camera.cpp:46-53 for code that wants single rays on the CPU.
#pragma once
#include <cmath>
#include <random>
#include "ray.h"

class Camera {
public:
    Vec3 m_oOrigin;
    Vec3 m_oLowerLeftCorner;
    Vec3 m_oHorizontal;
    Vec3 m_oVertical;
    Vec3 m_oU, m_oV, m_oW;
    float m_fLensRadius;

    Camera(Vec3 lookFrom, Vec3 lookAt, Vec3 up, float vfovDegrees, float aspect, float aperture, float focusDist)
        : m_oOrigin(lookFrom), m_fLensRadius(aperture / 2)
    {
        BuildFrame(lookAt, up);
        PlaceImagePlane(HalfExtentOfUnitPlane(vfovDegrees), aspect, focusDist);
    }

    // One sample of the unit disk by rejection from the unit square (camera.cpp:31-44); the generator is private to
    // the calling thread and seeded from the system, as in the reference.
    Vec3 RandomUnitInDisk() const
    {
        thread_local std::mt19937 engine{ std::random_device{}() };
        thread_local std::uniform_real_distribution<float> unit(0.0f, 1.0f);
        for (;;) {
            const float a = unit(engine);
            const float b = unit(engine);
            const Vec3 candidate = 2.0f * Vec3(a, b, 0.0f) - Vec3(1.0f, 1.0f, 0.0f);
            if (Dot(candidate, candidate) < 1.0f) return candidate;
        }
    }

    Ray GetRay(float s, float t) const
    {
        const Vec3 onLens = m_fLensRadius * RandomUnitInDisk();
        const Vec3 shift = m_oU * onLens.GetX() + m_oV * onLens.GetY();
        const Vec3 onFocusPlane = m_oLowerLeftCorner + s * m_oHorizontal + t * m_oVertical;
        return Ray(m_oOrigin + shift, onFocusPlane - m_oOrigin - shift);
    }

private:
    // tan(vfov / 2) with the angle converted in float and the float overload of tan, as the reference's
    // translation unit resolves it.
    static float HalfExtentOfUnitPlane(float vfovDegrees)
    {
        const float radians = vfovDegrees * static_cast<float>(3.14159265358979323846) / 180;
        return std::tan(radians / 2);
    }

    // w points from the target back to the eye, u to the right, v up.
    void BuildFrame(const Vec3& lookAt, const Vec3& up)
    {
        m_oW = Unit_Vector(m_oOrigin - lookAt);
        m_oU = Unit_Vector(Cross(up, m_oW));
        m_oV = Cross(m_oW, m_oU);
    }

    // The focus plane sits focusDist in front of the eye; its lower left corner and its two edge vectors.
    void PlaceImagePlane(float halfHeight, float aspect, float focusDist)
    {
        const float halfWidth = aspect * halfHeight;
        Vec3 corner = m_oOrigin - halfWidth * focusDist * m_oU;
        corner = corner - halfHeight * focusDist * m_oV;
        m_oLowerLeftCorner = corner - focusDist * m_oW;
        m_oHorizontal = 2 * halfWidth * focusDist * m_oU;
        m_oVertical = 2 * halfHeight * focusDist * m_oV;
    }
};
