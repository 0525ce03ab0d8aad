// A ray as the reference's host code knows it (ray.h:6-21): two public Vec3 members under the reference's
// names, origin and an un-normalised direction, plus the three accessors its callers use.
#pragma once
#include "vec3.h"

class Ray {
public:
    Vec3 m_oOrigin;
    Vec3 m_oDirection;

    Ray() = default;
    Ray(const Vec3& origin, const Vec3& direction) : m_oOrigin(origin), m_oDirection(direction) {}

    Vec3 Origin() const { return m_oOrigin; }
    Vec3 Direction() const { return m_oDirection; }

    // origin + t * direction; the name is spelled as the reference spells it, callers depend on it
    Vec3 PointAtParamenter(float t) const
    {
        const Vec3 along = t * m_oDirection;
        return m_oOrigin + along;
    }
};
