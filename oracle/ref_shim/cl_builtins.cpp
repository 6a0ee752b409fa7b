// cl_builtins.cpp -- TEST INFRASTRUCTURE (never linked into the product).
//
// The OpenCL 1.2 builtin functions that the reference's UNMODIFIED kernels
// (/root/reference/src/kernels/cl/*.cl, compiled for x86-64 by
// oracle/Makefile) leave undefined.  Symbol names are the Itanium manglings
// clang emits for OpenCL C (checked with nm); vector types are clang
// ext_vector_types so the by-value ABI matches the kernel objects.
//
// Definitions here are NORMATIVE for the project (DESIGN.md "arithmetic
// contract"): geometric builtins follow the OpenCL 1.2 spec formulas with a
// fixed left-to-right evaluation order; transcendentals come from
// raytracing_amd/csrc/rt_detmath.h (or glibc libm when built with
// -DREF_USE_LIBM, used only for the tolerance test).
#include <stdint.h>
#include <stddef.h>
#include "rt_detmath.h"

#ifdef REF_USE_LIBM
extern "C" {
float sinf(float); float cosf(float); float tanf(float); float powf(float, float);
float atan2f(float, float); float acosf(float); float ldexpf(float, int); float expf(float);
}
#define M_SIN(x) sinf(x)
#define M_COS(x) cosf(x)
#define M_TAN(x) tanf(x)
#define M_POW(x, y) powf(x, y)
#define M_ATAN2(y, x) atan2f(y, x)
#define M_ACOS(x) acosf(x)
#define M_LDEXP(x, k) ldexpf(x, k)
#define M_EXP(x) expf(x)
#else
#define M_SIN(x) rt_sinf(x)
#define M_COS(x) rt_cosf(x)
#define M_TAN(x) rt_tanf(x)
#define M_POW(x, y) rt_powf(x, y)
#define M_ATAN2(y, x) rt_atan2f(y, x)
#define M_ACOS(x) rt_acosf(x)
#define M_LDEXP(x, k) rt_ldexpf(x, k)
#define M_EXP(x) rt_expf(x)
#endif

typedef float float2 __attribute__((ext_vector_type(2)));
typedef float float3 __attribute__((ext_vector_type(3)));
typedef float float4 __attribute__((ext_vector_type(4)));
typedef int int2 __attribute__((ext_vector_type(2)));

// ---- work-item id ---------------------------------------------------------
extern "C" { thread_local size_t ref_global_id = 0; }
size_t shim_get_global_id(unsigned int) __asm__("_Z13get_global_idj");
size_t shim_get_global_id(unsigned int) { return ref_global_id; }

unsigned int shim_atomic_add(volatile unsigned int* p, unsigned int v) __asm__("_Z10atomic_addPU8CLglobalVjj");
unsigned int shim_atomic_add(volatile unsigned int* p, unsigned int v)
{
    return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
}

// ---- scalar ---------------------------------------------------------------
// OpenCL 1.2 6.12.4: min(x,y) = y < x ? y : x ; max(x,y) = x < y ? y : x
static inline float fmin_cl(float x, float y) { return y < x ? y : x; }
static inline float fmax_cl(float x, float y) { return x < y ? y : x; }

float shim_min(float x, float y) __asm__("_Z3minff");
float shim_min(float x, float y) { return fmin_cl(x, y); }
float shim_max(float x, float y) __asm__("_Z3maxff");
float shim_max(float x, float y) { return fmax_cl(x, y); }
float shim_fabs(float x) __asm__("_Z4fabsf");
float shim_fabs(float x) { return __builtin_fabsf(x); }
float shim_floor(float x) __asm__("_Z5floorf");
float shim_floor(float x) { return __builtin_floorf(x); }
float shim_sqrt(float x) __asm__("_Z4sqrtf");
float shim_sqrt(float x) { return __builtin_sqrtf(x); }
double shim_sqrtd(double x) __asm__("_Z4sqrtd");
double shim_sqrtd(double x) { return __builtin_sqrt(x); }
float shim_sin(float x) __asm__("_Z3sinf");
float shim_sin(float x) { return M_SIN(x); }
float shim_cos(float x) __asm__("_Z3cosf");
float shim_cos(float x) { return M_COS(x); }
float shim_tan(float x) __asm__("_Z3tanf");
float shim_tan(float x) { return M_TAN(x); }
float shim_exp(float x) __asm__("_Z3expf");
float shim_exp(float x) { return M_EXP(x); }
float shim_pow(float x, float y) __asm__("_Z3powff");
float shim_pow(float x, float y) { return M_POW(x, y); }
float shim_atan2(float y, float x) __asm__("_Z5atan2ff");
float shim_atan2(float y, float x) { return M_ATAN2(y, x); }
float shim_acos(float x) __asm__("_Z4acosf");
float shim_acos(float x) { return M_ACOS(x); }
float shim_ldexp(float x, int k) __asm__("_Z5ldexpfi");
float shim_ldexp(float x, int k) { return M_LDEXP(x, k); }
int shim_clampi(int x, int lo, int hi) __asm__("_Z5clampiii");
int shim_clampi(int x, int lo, int hi)
{
    // OpenCL 1.2 6.12.3: clamp(x, lo, hi) = min(max(x, lo), hi)
    int m = x < lo ? lo : x;
    return hi < m ? hi : m;
}

// ---- float3 ---------------------------------------------------------------
float shim_dot(float3 a, float3 b) __asm__("_Z3dotDv3_fS_");
float shim_dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

float3 shim_cross(float3 a, float3 b) __asm__("_Z5crossDv3_fS_");
float3 shim_cross(float3 a, float3 b)
{
    float3 r;
    r.x = a.y * b.z - a.z * b.y;
    r.y = a.z * b.x - a.x * b.z;
    r.z = a.x * b.y - a.y * b.x;
    return r;
}

float shim_length(float3 a) __asm__("_Z6lengthDv3_f");
float shim_length(float3 a) { return __builtin_sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }

float3 shim_normalize(float3 a) __asm__("_Z9normalizeDv3_f");
float3 shim_normalize(float3 a)
{
    float l = __builtin_sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
    float3 r;
    r.x = a.x / l;
    r.y = a.y / l;
    r.z = a.z / l;
    return r;
}

float3 shim_min3(float3 a, float3 b) __asm__("_Z3minDv3_fS_");
float3 shim_min3(float3 a, float3 b)
{
    float3 r;
    r.x = fmin_cl(a.x, b.x); r.y = fmin_cl(a.y, b.y); r.z = fmin_cl(a.z, b.z);
    return r;
}

float3 shim_max3(float3 a, float3 b) __asm__("_Z3maxDv3_fS_");
float3 shim_max3(float3 a, float3 b)
{
    float3 r;
    r.x = fmax_cl(a.x, b.x); r.y = fmax_cl(a.y, b.y); r.z = fmax_cl(a.z, b.z);
    return r;
}

// mix(x, y, a) = x + (y - x) * a   (OpenCL 1.2 6.12.4)
float3 shim_mix(float3 x, float3 y, float3 a) __asm__("_Z3mixDv3_fS_S_");
float3 shim_mix(float3 x, float3 y, float3 a)
{
    float3 r;
    r.x = x.x + (y.x - x.x) * a.x;
    r.y = x.y + (y.y - x.y) * a.y;
    r.z = x.z + (y.z - x.z) * a.z;
    return r;
}

float3 shim_mix_s(float3 x, float3 y, float a) __asm__("_Z3mixDv3_fS_f");
float3 shim_mix_s(float3 x, float3 y, float a)
{
    float3 r;
    r.x = x.x + (y.x - x.x) * a;
    r.y = x.y + (y.y - x.y) * a;
    r.z = x.z + (y.z - x.z) * a;
    return r;
}

float3 shim_pow3(float3 x, float3 y) __asm__("_Z3powDv3_fS_");
float3 shim_pow3(float3 x, float3 y)
{
    float3 r;
    r.x = M_POW(x.x, y.x); r.y = M_POW(x.y, y.y); r.z = M_POW(x.z, y.z);
    return r;
}

float3 shim_clamp3(float3 x, float lo, float hi) __asm__("_Z5clampDv3_fff");
float3 shim_clamp3(float3 x, float lo, float hi)
{
    float3 r;
    r.x = fmin_cl(fmax_cl(x.x, lo), hi);
    r.y = fmin_cl(fmax_cl(x.y, lo), hi);
    r.z = fmin_cl(fmax_cl(x.z, lo), hi);
    return r;
}

float2 shim_floor2(float2 x) __asm__("_Z5floorDv2_f");
float2 shim_floor2(float2 x)
{
    float2 r;
    r.x = __builtin_floorf(x.x); r.y = __builtin_floorf(x.y);
    return r;
}

// ---- images ---------------------------------------------------------------
// image2d_t arrives as an opaque pointer; the driver passes a ShimImage*.
struct ShimImage
{
    int width;
    int height;
    float* data;   // RGBA32F, row-major
};

extern "C" void* __translate_sampler_initializer(int v) { return (void*)(intptr_t)v; }

// OpenCL 1.2 spec 8.2 (CLK_NORMALIZED_COORDS_TRUE | CLK_ADDRESS_REPEAT |
// CLK_FILTER_LINEAR), the only sampler the path uses (miss.cl:30).
// Evaluation order fixed here; mirrored by oracle.c and raytracing_amd/csrc/device_math.h.
float4 shim_read_imagef(ShimImage* img, void* sampler, float2 coord)
    __asm__("_Z11read_imagef14ocl_image2d_ro11ocl_samplerDv2_f");
float4 shim_read_imagef(ShimImage* img, void* /*sampler*/, float2 coord)
{
    int w = img->width, h = img->height;
    float u = (coord.x - __builtin_floorf(coord.x)) * (float)w;
    float v = (coord.y - __builtin_floorf(coord.y)) * (float)h;
    float fu = __builtin_floorf(u - 0.5f);
    float fv = __builtin_floorf(v - 0.5f);
    // OpenCL leaves read_imagef undefined for NaN coordinates (6.12.14); they DO occur: acos of a direction component one ulp
    // above 1 (miss.cl:34), once in ~1e9 escaped rays.  x86 converts NaN to INT_MIN (an address far outside the image: the
    // reference's kernels crashed here in a 128-spp run of the config-5 stand-in), gfx950 to 0.  Defined for the project: texel
    // (0, 0), and the NaN weights make the sample NaN -- what the GPU path always produced.
    int i0 = fu != fu ? 0 : (int)fu;
    int j0 = fv != fv ? 0 : (int)fv;
    int i1 = i0 + 1;
    int j1 = j0 + 1;
    if (i0 < 0) i0 = w + i0;
    if (i1 > w - 1) i1 = i1 - w;
    if (j0 < 0) j0 = h + j0;
    if (j1 > h - 1) j1 = j1 - h;
    float a = (u - 0.5f) - fu;
    float b = (v - 0.5f) - fv;
    float wa0 = 1.0f - a;
    float wb0 = 1.0f - b;
    const float* t00 = img->data + 4 * ((size_t)j0 * w + i0);
    const float* t10 = img->data + 4 * ((size_t)j0 * w + i1);
    const float* t01 = img->data + 4 * ((size_t)j1 * w + i0);
    const float* t11 = img->data + 4 * ((size_t)j1 * w + i1);
    float w00 = wa0 * wb0, w10 = a * wb0, w01 = wa0 * b, w11 = a * b;
    float4 r;
    r.x = w00 * t00[0] + w10 * t10[0] + w01 * t01[0] + w11 * t11[0];
    r.y = w00 * t00[1] + w10 * t10[1] + w01 * t01[1] + w11 * t11[1];
    r.z = w00 * t00[2] + w10 * t10[2] + w01 * t01[2] + w11 * t11[2];
    r.w = w00 * t00[3] + w10 * t10[3] + w01 * t01[3] + w11 * t11[3];
    return r;
}

void shim_write_imagef(ShimImage* img, int2 coord, float4 color)
    __asm__("_Z12write_imagef14ocl_image2d_woDv2_iDv4_f");
void shim_write_imagef(ShimImage* img, int2 coord, float4 color)
{
    float* p = img->data + 4 * ((size_t)coord.y * img->width + coord.x);
    p[0] = color.x; p[1] = color.y; p[2] = color.z; p[3] = color.w;
}
