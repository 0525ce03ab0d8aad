// Host mirror of the reference's Vec3 (vec3.h:8-135, vec3.cpp): same public member m_fAxis[3], same
// accessors, operators and free functions, same arithmetic (operator/(Vec3,float) divides,
// operator/=(float) multiplies by the reciprocal).  Header-only; nothing here runs on the GPU.
#pragma once
#include <cmath>
#include <iosfwd>
#include <istream>
#include <ostream>

class Vec3 {
public:
    float m_fAxis[3];

    Vec3() {}
    Vec3(const float& e0, const float& e1, const float& e2) : m_fAxis{ e0, e1, e2 } {}

    float GetX() const { return m_fAxis[0]; }
    float GetY() const { return m_fAxis[1]; }
    float GetZ() const { return m_fAxis[2]; }
    float GetR() const { return m_fAxis[0]; }
    float GetG() const { return m_fAxis[1]; }
    float GetB() const { return m_fAxis[2]; }

    const Vec3& operator+() const { return *this; }
    Vec3 operator-() const { return Vec3(-m_fAxis[0], -m_fAxis[1], -m_fAxis[2]); }
    float operator[](int i) const { return m_fAxis[i]; }
    float& operator[](int i) { return m_fAxis[i]; }

    Vec3& operator+=(const Vec3& r) { for (int k = 0; k < 3; ++k) m_fAxis[k] += r.m_fAxis[k]; return *this; }
    Vec3& operator-=(const Vec3& r) { for (int k = 0; k < 3; ++k) m_fAxis[k] -= r.m_fAxis[k]; return *this; }
    Vec3& operator*=(const Vec3& r) { for (int k = 0; k < 3; ++k) m_fAxis[k] *= r.m_fAxis[k]; return *this; }
    Vec3& operator/=(const Vec3& r) { for (int k = 0; k < 3; ++k) m_fAxis[k] /= r.m_fAxis[k]; return *this; }
    Vec3& operator*=(float s) { for (int k = 0; k < 3; ++k) m_fAxis[k] *= s; return *this; }
    Vec3& operator/=(float s) { const float inv = 1.0f / s; for (int k = 0; k < 3; ++k) m_fAxis[k] *= inv; return *this; }

    float SquaredLength() const { return m_fAxis[0] * m_fAxis[0] + m_fAxis[1] * m_fAxis[1] + m_fAxis[2] * m_fAxis[2]; }
    float Length() const { return sqrtf(SquaredLength()); }
    void MakeUnitVector() { const float inv = 1.0f / Length(); for (int k = 0; k < 3; ++k) m_fAxis[k] *= inv; }
};

inline std::istream& operator>>(std::istream& in, Vec3& v) { return in >> v.m_fAxis[0] >> v.m_fAxis[1] >> v.m_fAxis[2]; }
inline std::ostream& operator<<(std::ostream& out, const Vec3& v) { return out << "X: " << v.m_fAxis[0] << "Y: " << v.m_fAxis[1] << "Z: " << v.m_fAxis[2]; }

#define RT_VEC3_BINOP(OP) \
    inline Vec3 operator OP(const Vec3& a, const Vec3& b) { return Vec3(a.m_fAxis[0] OP b.m_fAxis[0], a.m_fAxis[1] OP b.m_fAxis[1], a.m_fAxis[2] OP b.m_fAxis[2]); }
RT_VEC3_BINOP(+)
RT_VEC3_BINOP(-)
RT_VEC3_BINOP(*)
RT_VEC3_BINOP(/)
#undef RT_VEC3_BINOP
inline Vec3 operator*(float s, const Vec3& v) { return Vec3(s * v.m_fAxis[0], s * v.m_fAxis[1], s * v.m_fAxis[2]); }
inline Vec3 operator*(const Vec3& v, float s) { return Vec3(v.m_fAxis[0] * s, v.m_fAxis[1] * s, v.m_fAxis[2] * s); }
inline Vec3 operator/(const Vec3& v, float s) { return Vec3(v.m_fAxis[0] / s, v.m_fAxis[1] / s, v.m_fAxis[2] / s); }

inline float Dot(const Vec3& a, const Vec3& b) { return (a.m_fAxis[0] * b.m_fAxis[0]) + (a.m_fAxis[1] * b.m_fAxis[1]) + (a.m_fAxis[2] * b.m_fAxis[2]); }
inline Vec3 Cross(const Vec3& a, const Vec3& b)
{
    return Vec3(a.m_fAxis[1] * b.m_fAxis[2] - a.m_fAxis[2] * b.m_fAxis[1],
                -(a.m_fAxis[0] * b.m_fAxis[2] - a.m_fAxis[2] * b.m_fAxis[0]),
                a.m_fAxis[0] * b.m_fAxis[1] - a.m_fAxis[1] * b.m_fAxis[0]);
}
inline Vec3 Unit_Vector(Vec3 v) { return v / v.Length(); }
