// Host mirror of the reference's Ray (ray.h:6-21, ray.cpp): origin plus un-normalised direction.
#pragma once
#include "vec3.h"

class Ray {
public:
    Vec3 m_oOrigin;
    Vec3 m_oDirection;

    Ray() {}
    Ray(const Vec3& a, const Vec3& b) : m_oOrigin(a), m_oDirection(b) {}
    Vec3 Origin() const { return m_oOrigin; }
    Vec3 Direction() const { return m_oDirection; }
    Vec3 PointAtParamenter(float t) const { return m_oOrigin + t * m_oDirection; }   // spelling as in the reference
};
