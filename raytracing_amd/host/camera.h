// Host mirror of the reference's Camera (camera.h:6-22, camera.cpp:6-53): same constructor, same
// eight public fields.  GetRay() keeps a CPU implementation for host code; the render loop reads the
// fields through rt_camera_pod instead.
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
    {
        m_fLensRadius = aperture / 2;
        const float theta = vfovDegrees * static_cast<float>(3.14159265358979323846) / 180;
        const float halfHeight = std::tan(theta / 2);            // float overload, as in the reference's TU
        const float halfWidth = aspect * halfHeight;
        m_oOrigin = lookFrom;
        m_oW = Unit_Vector(lookFrom - lookAt);
        m_oU = Unit_Vector(Cross(up, m_oW));
        m_oV = Cross(m_oW, m_oU);
        m_oLowerLeftCorner = m_oOrigin - halfWidth * focusDist * m_oU - halfHeight * focusDist * m_oV - focusDist * m_oW;
        m_oHorizontal = 2 * halfWidth * focusDist * m_oU;
        m_oVertical = 2 * halfHeight * focusDist * m_oV;
    }

    Vec3 RandomUnitInDisk() const
    {
        thread_local std::mt19937 gen{ std::random_device{}() };
        thread_local std::uniform_real_distribution<float> dis(0.0f, 1.0f);
        Vec3 p;
        do {
            const float x = dis(gen), y = dis(gen);
            p = 2.0f * Vec3(x, y, 0.0f) - Vec3(1.0f, 1.0f, 0.0f);
        } while (Dot(p, p) >= 1.0f);
        return p;
    }

    Ray GetRay(float s, float t) const
    {
        const Vec3 rd = m_fLensRadius * RandomUnitInDisk();
        const Vec3 offset = m_oU * rd.GetX() + m_oV * rd.GetY();
        return Ray(m_oOrigin + offset, m_oLowerLeftCorner + s * m_oHorizontal + t * m_oVertical - m_oOrigin - offset);
    }
};
