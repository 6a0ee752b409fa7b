// device_math.h -- fp32 vector helpers for the gfx950 kernels.
//
// ARITHMETIC CONTRACT: every expression in the kernels is written
// with the evaluation order of the reference's OpenCL source, the OpenCL
// geometric builtins are expanded to the project-normative formulas (dot and
// length summed left to right, normalize = three IEEE divides by the length,
// mix = x + (y - x) * a, min/max = the OpenCL 1.2 select forms), and the whole
// translation unit is compiled with -ffp-contract=off, correctly rounded
// divide/sqrt and without fast-math, so results are bit-identical to the CPU
// oracle.  Do NOT "optimise" these into rcp/rsq/fma forms.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rt_detmath.h"

#define RT_DEV __device__ __forceinline__

struct f3 { float x, y, z; };
struct f2 { float x, y; };

RT_DEV f3 F3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_DEV f3 F3s(float s) { return F3(s, s, s); }
RT_DEV f3 xyz(float4 v) { return F3(v.x, v.y, v.z); }

// OpenCL 1.2 6.12.4: min(x, y) = y < x ? y : x ; max(x, y) = x < y ? y : x
RT_DEV float cl_min(float x, float y) { return y < x ? y : x; }
RT_DEV float cl_max(float x, float y) { return x < y ? y : x; }
RT_DEV int cl_clampi(int x, int lo, int hi) { int m = x < lo ? lo : x; return hi < m ? hi : m; }

RT_DEV f3 operator+(f3 a, f3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV f3 operator-(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV f3 operator*(f3 a, f3 b) { return F3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_DEV f3 operator*(f3 a, float s) { return F3(a.x * s, a.y * s, a.z * s); }
RT_DEV f3 operator/(f3 a, float s) { return F3(a.x / s, a.y / s, a.z / s); }
RT_DEV f3 operator-(f3 a) { return F3(-a.x, -a.y, -a.z); }

RT_DEV float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_DEV f3 cross3(f3 a, f3 b)
{
    return F3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
RT_DEV float length3(f3 a) { return __builtin_sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
RT_DEV f3 normalize3(f3 a)
{
    float l = __builtin_sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
    return F3(a.x / l, a.y / l, a.z / l);
}
RT_DEV f3 mix3(f3 x, f3 y, float a)
{
    return F3(x.x + (y.x - x.x) * a, x.y + (y.y - x.y) * a, x.z + (y.z - x.z) * a);
}

// utils.h:113-121
RT_DEV uint32_t WangHash(uint32_t x)
{
    x = (x ^ 61u) ^ (x >> 16);
    x = x + (x << 3);
    x = x ^ (x >> 4);
    x = x * 0x27d4eb2du;
    x = x ^ (x >> 15);
    return x;
}

// raygeneration.cl:28-38
RT_DEV float GetRandomFloat(uint32_t& seed)
{
    uint32_t s = WangHash(seed);
    s = 1103515245u * s + 12345u;
    seed = s;
    return (float)s * 2.3283064365386963e-10f;
}

// sampling.h:64-82, kRandom.  The first two hash levels depend on the pixel
// only and are hoisted by the caller (pixel_seed); the value is unchanged.
RT_DEV uint32_t SampleRandomPixelSeed(uint32_t px, uint32_t py)
{
    uint32_t seed = WangHash(px);
    return WangHash(seed + WangHash(py));
}
RT_DEV uint32_t SampleRandomSampleSeed(uint32_t pixel_seed, uint32_t sample_index)
{
    return WangHash(pixel_seed + WangHash(sample_index));
}
RT_DEV float SampleRandomDim(uint32_t sample_seed, uint32_t bounce, uint32_t type)
{
    uint32_t dim = bounce * 5u + type;
    uint32_t seed = WangHash(sample_seed + WangHash(dim));
    return (float)seed * 2.3283064365386963e-10f;
}
