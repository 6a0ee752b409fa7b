// mathlib.hpp -- host-side vector/bounds types of the HIP backend's host layer.
// Same value semantics as the reference's src/mathlib/mathlib.hpp:40-230 (16-byte
// float3 with explicit pad so host arrays match the device records, Bounds3 with
// FLT_MAX/lowest empty state, Union/Offset/SurfaceArea/MaximumExtent) because
// the BVH builder's decisions depend on them bit for bit.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>

namespace rt
{
struct float3
{
    float x = 0.0f, y = 0.0f, z = 0.0f;
    std::uint32_t pad = 0;

    float3() = default;
    float3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    explicit float3(float v) : x(v), y(v), z(v) {}

    float Length() const { return std::sqrt(x * x + y * y + z * z); }
    float3 Normalize() const { return float3(x / Length(), y / Length(), z / Length()); }
    float operator[](std::size_t i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float& operator[](std::size_t i) { return i == 0 ? x : (i == 1 ? y : z); }
};
static_assert(sizeof(float3) == 16, "float3 must match the 16-byte device float3");

struct float2
{
    float x = 0.0f, y = 0.0f;
    float2() = default;
    float2(float x_, float y_) : x(x_), y(y_) {}
};

inline float3 operator+(const float3& a, const float3& b) { return float3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline float3 operator-(const float3& a, const float3& b) { return float3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline float3 operator*(const float3& a, float s) { return float3(a.x * s, a.y * s, a.z * s); }
inline float3 Cross(const float3& a, const float3& b)
{
    return float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float Dot(const float3& a, const float3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float3 Min(const float3& a, const float3& b)
{
    return float3(std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z));
}
inline float3 Max(const float3& a, const float3& b)
{
    return float3(std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z));
}

struct Bounds3
{
    float3 min, max;

    Bounds3()
    {
        const float hi = std::numeric_limits<float>::max();
        const float lo = std::numeric_limits<float>::lowest();
        min = float3(hi, hi, hi);
        max = float3(lo, lo, lo);
    }
    explicit Bounds3(const float3& p) : min(p), max(p) {}
    Bounds3(const float3& a, const float3& b) : min(Min(a, b)), max(Max(a, b)) {}

    float3 Diagonal() const { return max - min; }
    float SurfaceArea() const
    {
        float3 d = Diagonal();
        return 2 * (d.x * d.y + d.x * d.z + d.y * d.z);
    }
    unsigned MaximumExtent() const
    {
        float3 d = Diagonal();
        if (d.x > d.y && d.x > d.z) return 0;
        return d.y > d.z ? 1u : 2u;
    }
    float3 Offset(const float3& p) const
    {
        float3 o = p - min;
        if (max.x > min.x) o.x /= max.x - min.x;
        if (max.y > min.y) o.y /= max.y - min.y;
        if (max.z > min.z) o.z /= max.z - min.z;
        return o;
    }
};
static_assert(sizeof(Bounds3) == 32, "Bounds3");

inline Bounds3 Union(const Bounds3& b, const float3& p)
{
    Bounds3 r;
    r.min = Min(b.min, p);
    r.max = Max(b.max, p);
    return r;
}
inline Bounds3 Union(const Bounds3& a, const Bounds3& b)
{
    Bounds3 r;
    r.min = Min(a.min, b.min);
    r.max = Max(a.max, b.max);
    return r;
}

template <class V>
inline V clamp(V v, V lo, V hi) { return v < lo ? lo : (v > hi ? hi : v); }
} // namespace rt
