/* rt_types.h -- plain-C restatement of the host<->device POD records the
 * reference shares between its C++ host and its OpenCL kernels
 * (reference: src/kernels/common/shared_structures.h:56-181, host float3 with
 * explicit pad src/mathlib/mathlib.hpp:40-77).  Sizes/offsets are asserted
 * below and again against the compiled reference in tests/test_ref_pin.py.
 *
 * These are the records that cross the C-ABI in include/rt_hip.h; the HIP
 * backend re-lays them out for the device behind that boundary.
 */
#ifndef RT_TYPES_H
#define RT_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_MAX_RENDER_DIST 20000.0f   /* constants.h:28 */
#define RT_EPS 1e-3f                  /* constants.h:29 */
#define RT_PI 3.14159265359f          /* constants.h:30 */
#define RT_TWO_PI 6.28318530718f      /* constants.h:31 */
#define RT_INV_PI 0.31830988618f      /* constants.h:32 */
#define RT_INV_TWO_PI 0.15915494309f  /* constants.h:33 */
#define RT_INVALID_ID 0xFFFFFFFFu     /* constants.h:34 */
#define RT_INVALID_TEXTURE_IDX 0xFFu  /* constants.h:35 */

#define RT_LIGHT_TYPE_POINT 0u        /* shared_structures.h:39 */
#define RT_LIGHT_TYPE_DIRECTIONAL 1u  /* shared_structures.h:40 */

/* 16-byte float3: x,y,z + pad (OpenCL float3 / mathlib.hpp:73-76) */
typedef struct rt_float3 { float x, y, z, w; } rt_float3;
typedef struct rt_float4 { float x, y, z, w; } rt_float4;
typedef struct rt_float2 { float x, y; } rt_float2;

typedef struct rt_ray            /* shared_structures.h:56-59 */
{
    rt_float4 origin;            /* w = t_min */
    rt_float4 direction;         /* w = t_max */
} rt_ray;

typedef struct rt_hit            /* shared_structures.h:61-66 */
{
    rt_float2 bc;
    uint32_t primitive_id;
    float t;
} rt_hit;

typedef struct rt_scene_info     /* shared_structures.h:68-73 */
{
    uint32_t analytic_light_count;
    uint32_t emissive_count;
    uint32_t environment_map_index;
    uint32_t padding;
} rt_scene_info;

typedef struct rt_packed_material /* shared_structures.h:75-81 */
{
    uint32_t diffuse_albedo;
    uint32_t specular_albedo;
    uint32_t emission;
    uint32_t roughness_metalness;
    uint32_t ior_emission_idx_transparency;
} rt_packed_material;

typedef struct rt_light          /* shared_structures.h:83-88 */
{
    rt_float3 origin;
    rt_float3 radiance;
    uint32_t type;
    uint32_t padding[3];
} rt_light;

typedef struct rt_texture        /* shared_structures.h:90-95 */
{
    int32_t data_start;
    int32_t width;
    int32_t height;
    int32_t padding;
} rt_texture;

typedef struct rt_vertex         /* shared_structures.h:97-109 */
{
    rt_float3 position;
    rt_float3 texcoord;
    rt_float3 normal;
} rt_vertex;

typedef struct rt_triangle       /* shared_structures.h:111-141 */
{
    rt_vertex v1, v2, v3;
    uint32_t mtl_index;
    uint32_t padding[3];
} rt_triangle;

typedef struct rt_bvh_node       /* LinearBVHNode, shared_structures.h:160-171 */
{
    rt_float3 bounds_min;
    rt_float3 bounds_max;
    uint32_t offset;              /* first primitive (leaf) or second child */
    uint32_t num_primitives_axis; /* (n << 16) | axis ; n == 0 -> interior  */
    uint32_t padding[2];
} rt_bvh_node;

typedef struct rt_camera         /* shared_structures.h:173-181 */
{
    rt_float3 position;
    rt_float3 front;
    rt_float3 up;
    float fov;
    float aspect_ratio;
    float aperture;
    float focus_distance;
} rt_camera;

#ifdef __cplusplus
}
#define RT_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define RT_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif

RT_STATIC_ASSERT(sizeof(rt_ray) == 32, "Ray");
RT_STATIC_ASSERT(sizeof(rt_hit) == 16, "Hit");
RT_STATIC_ASSERT(sizeof(rt_scene_info) == 16, "SceneInfo");
RT_STATIC_ASSERT(sizeof(rt_packed_material) == 20, "PackedMaterial");
RT_STATIC_ASSERT(sizeof(rt_light) == 48, "Light");
RT_STATIC_ASSERT(sizeof(rt_texture) == 16, "Texture");
RT_STATIC_ASSERT(sizeof(rt_vertex) == 48, "Vertex");
RT_STATIC_ASSERT(sizeof(rt_triangle) == 160, "Triangle");
RT_STATIC_ASSERT(sizeof(rt_bvh_node) == 48, "LinearBVHNode");
RT_STATIC_ASSERT(sizeof(rt_camera) == 64, "Camera");

#endif /* RT_TYPES_H */
