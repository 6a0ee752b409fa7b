/* oracle.c -- TEST INFRASTRUCTURE ONLY.  Never linked into, imported by or
 * called from the product path (raytracing_amd/); only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() use it, as the checker.
 *
 * A plain-C, single-threaded restatement of the reference's wavefront path
 * tracing loop, one function per reference kernel, each citing the file:line
 * it follows (paths relative to /root/reference).  Arithmetic is restated
 * operation for operation (evaluation order, fp32/fp64 mix, IEEE NaN paths);
 * the OpenCL builtins are the project-normative definitions of
 * oracle/ref_shim/cl_builtins.cpp + raytracing_amd/csrc/rt_detmath.h.
 *
 * PINNING: tests/test_ref_pin.py checks this file BIT FOR BIT, stage by stage
 * and end to end, against oracle/_ref/libref.so = the reference's own
 * unmodified .cl kernels compiled for x86-64 (the reference ships no golden
 * vectors or tests of its own), and against the golden
 * fixtures under tests/golden/ that were generated from that build.
 *
 * Queue order: appends are sequential in work-item order, which is exactly the
 * order the single-threaded reference build produces with its atomic_add
 * appends (hit_surface.cl:138,173), so intermediate buffers compare directly.
 *
 * Extra (not in the reference): traversal counters n_nodes / n_tris per ray
 * class -- the "algorithmic bytes" inputs of SURVEY.md section 8(d); the opt-in
 * extensions of rt_scene_desc (marked where they appear); and, at the end of the
 * file, orc_wide_trace: the HIP path's own 4-wide quantized walk (k_trace_w4)
 * restated ray by ray, so that tests/test_wide_traversal_oracle.py can pin THAT
 * algorithm to TraceOne below (= trace_bvh.cl:99-211) on the CPU.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "rt_types.h"
#include "rt_detmath.h"

typedef struct { float x, y, z; } v3;
typedef struct { float x, y; } v2;

/* ---- normative OpenCL builtins (cl_builtins.cpp) ------------------------ */
static inline float f_min(float x, float y) { return y < x ? y : x; }
static inline float f_max(float x, float y) { return x < y ? y : x; }
static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v_add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v_sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v_mul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v_scale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 v_divs(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline v3 v_neg(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline float v_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 v_cross(v3 a, v3 b)
{
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float v_length(v3 a) { return __builtin_sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
static inline v3 v_normalize(v3 a)
{
    float l = __builtin_sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
    return V3(a.x / l, a.y / l, a.z / l);
}
static inline v3 v_mix(v3 x, v3 y, float a)
{
    return V3(x.x + (y.x - x.x) * a, x.y + (y.y - x.y) * a, x.z + (y.z - x.z) * a);
}
static inline v3 f3(rt_float3 a) { return V3(a.x, a.y, a.z); }
static inline int i_clamp(int x, int lo, int hi) { int m = x < lo ? lo : x; return hi < m ? hi : m; }

/* ---- state -------------------------------------------------------------- */
typedef struct orc
{
    uint32_t width, height;
    int furnace;
    uint32_t max_bounces;
    int request_reset;
    rt_camera camera;
    rt_camera prev_camera;       /* Integrator::prev_camera_ */
    rt_camera aov_prev_camera;   /* kPrevCamera bound by the last SetCameraData (cl_pt_integrator.cpp:365-371) */
    int denoiser;
    int blue_noise;              /* Integrator::sampler_type_ == kBlueNoise */
    uint32_t aov;
    rt_scene_info scene_info;
    rt_float3* diffuse_albedo; float* depth; rt_float3* normal; rt_float2* velocity;
    rt_float4* prev_radiance; float* prev_depth;

    rt_float4* radiance;
    rt_ray* rays[2];
    uint32_t* pixel_indices[2];
    uint32_t ray_counter[2];
    rt_ray* shadow_rays;
    uint32_t* shadow_pixel_indices;
    uint32_t shadow_ray_counter;
    rt_hit* hits;
    uint32_t* shadow_hits;
    rt_float3* throughputs;
    uint32_t sample_counter;
    rt_float4* direct_light_samples;
    rt_float4* resolved;

    rt_triangle* triangles; uint32_t n_triangles;
    rt_bvh_node* nodes; uint32_t n_nodes;
    rt_packed_material* materials; uint32_t n_materials;
    rt_texture* textures; uint32_t n_textures;
    uint32_t* texture_data; uint32_t n_texture_data;
    rt_light* lights; uint32_t n_lights;
    float* env; uint32_t env_w, env_h;
    /* opt-in extensions of this repository (include/rt_hip.h, rt_scene_desc) -- NOT reference behaviour,
     * restated here so that the HIP path can be checked against something: */
    uint16_t* tex16;             /* 6 texture indices per material (0xFFFF = none) replacing the packed 8-bit ones, or NULL */
    uint32_t* emissive; uint32_t n_emissive_tris;
    int emissive_nee;            /* RT_SCENE_EMISSIVE_NEE */
    uint8_t* prev_delta;         /* per pixel: the path's last scattering event was a delta one (or it is a camera ray) */

    /* statistics */
    uint64_t total_closest, total_shadow;
    uint64_t closest_nodes, closest_tris, shadow_nodes, shadow_tris;
    uint64_t n_escaped, n_hits, n_emissive, n_texels, n_unoccluded, n_outgoing;
    uint32_t last_active[64], last_shadow[64];
} orc;

/* ---- RNG / hashing ------------------------------------------------------ */
/* utils.h:113-121 */
static uint32_t WangHash(uint32_t x)
{
    x = (x ^ 61u) ^ (x >> 16);
    x = x + (x << 3);
    x = x ^ (x >> 4);
    x = x * 0x27d4eb2du;
    x = x ^ (x >> 15);
    return x;
}

/* raygeneration.cl:28-38 */
static float GetRandomFloat(uint32_t* seed)
{
    uint32_t s = *seed;
    s = (s ^ 61u) ^ (s >> 16);
    s = s + (s << 3);
    s = s ^ (s >> 4);
    s = s * 0x27d4eb2du;
    s = s ^ (s >> 15);
    s = 1103515245u * s + 12345u;
    *seed = s;
    return (float)s * 2.3283064365386963e-10f;
}

/* SampleBlueNoise, sampling.h:40-61.  bn = sobol_256spp_256d[65536] | scramblingTile[131072] |
   rankingTile[131072 + 256 zero padding]: the reference indexes rankingTile with the un-wrapped
   dimension (:50), which runs past the table for the last pixels; the padding reads as 0. */
static const int* g_bn_sobol; static const int* g_bn_scramble; static const int* g_bn_rank;
static float SampleBlueNoise(int pixel_i, int pixel_j, int sampleIndex, int sampleDimension)
{
    pixel_i = pixel_i & 127;
    pixel_j = pixel_j & 127;
    sampleIndex = sampleIndex & 255;
    sampleDimension = sampleDimension & 255;
    int rankedSampleIndex = sampleIndex ^ g_bn_rank[sampleDimension + (pixel_i + pixel_j * 128) * 8];
    int value = g_bn_sobol[sampleDimension + rankedSampleIndex * 256];
    value = value ^ g_bn_scramble[(sampleDimension % 8) + (pixel_i + pixel_j * 128) * 8];
    float v = (0.5f + value) / 256.0f;
    return v;
}

/* sampling.h:64-82 */
static int g_blue_noise;
static float SampleRandom(uint32_t px, uint32_t py, uint32_t sample_index, uint32_t bounce, uint32_t type)
{
    uint32_t dim = bounce * 5u + type;
    if (g_blue_noise) return SampleBlueNoise((int)px, (int)py, (int)sample_index, (int)dim);
    uint32_t seed = WangHash(px);
    seed = WangHash(seed + WangHash(py));
    seed = WangHash(seed + WangHash(sample_index));
    seed = WangHash(seed + WangHash(dim));
    return (float)seed * 2.3283064365386963e-10f;
}

/* ---- RayGeneration, raygeneration.cl:65-139 ----------------------------- */
static void RayGeneration(orc* o, uint32_t ray_idx)
{
    uint32_t width = o->width, height = o->height;
    if (ray_idx >= width * height) return;
    const rt_camera* cam = &o->camera;
    uint32_t pixel_idx = ray_idx;
    uint32_t pixel_x = pixel_idx % width;
    uint32_t pixel_y = pixel_idx / width;
    float inv_width = 1.0f / (float)width;
    float inv_height = 1.0f / (float)height;
    uint32_t sample_idx = o->sample_counter;
    uint32_t seed = pixel_idx + (1103515245u * sample_idx + 12345u);      /* :61,98 */

    float x = ((float)pixel_x + GetRandomFloat(&seed)) * inv_width;        /* :101 */
    float y = ((float)pixel_y + GetRandomFloat(&seed)) * inv_height;       /* :102 */

    float angle = rt_tanf(0.5f * cam->fov);                                /* :108 */
    x = (x * 2.0f - 1.0f) * angle * cam->aspect_ratio;
    y = (y * 2.0f - 1.0f) * angle;

    v3 front = f3(cam->front), up = f3(cam->up), pos = f3(cam->position);
    v3 right = v_cross(front, up);
    v3 dir = v_normalize(v_add(v_add(v_scale(right, x), v_scale(up, y)), front));   /* :112 */

    v3 point_aimed = v_add(pos, v_scale(dir, cam->focus_distance));       /* :115 */
    /* PointInHexagon :40-49 */
    static const float hx[4] = {-1.0f, 0.5f, 0.5f, 0.0f};
    static const float hy[4] = {0.0f, 0.866f, -0.866f, 0.0f};
    int hidx = (int)__builtin_floorf(GetRandomFloat(&seed) * 3.0f);
    /* the reference indexes hexPoints[3] (out of bounds, UB) when the draw is
       exactly 1.0f (p ~ 3e-8); this restatement defines that entry as (0,0) */
    int h1 = hidx > 3 ? 3 : hidx;
    int h2 = (hidx + 1) % 3;
    float p1 = GetRandomFloat(&seed);
    float p2 = GetRandomFloat(&seed);
    float dofx = p1 * hx[h1] + p2 * hx[h2];
    float dofy = p1 * hy[h1] + p2 * hy[h2];
    float r = cam->aperture;
    v3 new_pos = v_add(v_add(pos, v_scale(right, dofx * r)), v_scale(up, dofy * r));  /* :118 */

    rt_ray ray;
    ray.origin.x = new_pos.x; ray.origin.y = new_pos.y; ray.origin.z = new_pos.z; ray.origin.w = 0.0f;
    v3 d = v_normalize(v_sub(point_aimed, new_pos));
    ray.direction.x = d.x; ray.direction.y = d.y; ray.direction.z = d.z;
    ray.direction.w = RT_MAX_RENDER_DIST;

    o->rays[0][ray_idx] = ray;
    o->pixel_indices[0][ray_idx] = pixel_idx;
    o->throughputs[pixel_idx].x = 1.0f;
    o->throughputs[pixel_idx].y = 1.0f;
    o->throughputs[pixel_idx].z = 1.0f;
    o->diffuse_albedo[pixel_idx].x = 0.0f; o->diffuse_albedo[pixel_idx].y = 0.0f; o->diffuse_albedo[pixel_idx].z = 0.0f;
    o->depth[pixel_idx] = RT_MAX_RENDER_DIST;                             /* :129-132 */
    o->normal[pixel_idx].x = 0.0f; o->normal[pixel_idx].y = 0.0f; o->normal[pixel_idx].z = 0.0f;
    o->velocity[pixel_idx].x = 0.0f; o->velocity[pixel_idx].y = 0.0f;
    if (ray_idx == 0) o->ray_counter[0] = width * height;
}

/* ---- TraceBvh, trace_bvh.cl:28-211 -------------------------------------- */
/* RayTriangle :28-73 */
static int RayTriangle(v3 org, v3 dir, float t_min, float t_max, const rt_triangle* tri, float* bu, float* bv,
    float* out_t)
{
    v3 p1 = f3(tri->v1.position), p2 = f3(tri->v2.position), p3 = f3(tri->v3.position);
    v3 e1 = v_sub(p2, p1);
    v3 e2 = v_sub(p3, p1);
    v3 pvec = v_cross(dir, e2);
    float det = v_dot(e1, pvec);
    if (det < 1e-8f || -det > 1e-8f) return 0;
    float inv_det = 1.0f / det;
    v3 tvec = v_sub(org, p1);
    float u = v_dot(tvec, pvec) * inv_det;
    if (u < 0.0f || u > 1.0f) return 0;
    v3 qvec = v_cross(tvec, e1);
    float v = v_dot(dir, qvec) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return 0;
    float t = v_dot(e2, qvec) * inv_det;
    if (t < t_min || t > t_max) return 0;
    *bu = u; *bv = v; *out_t = t;
    return 1;
}

/* RayBounds :85-97 */
static int RayBounds(const rt_bvh_node* n, v3 org, v3 inv, float t_min, float t_max)
{
    v3 t0 = v_mul(v_sub(f3(n->bounds_min), org), inv);
    v3 t1 = v_mul(v_sub(f3(n->bounds_max), org), inv);
    v3 lo = V3(f_min(t0.x, t1.x), f_min(t0.y, t1.y), f_min(t0.z, t1.z));
    v3 hi = V3(f_max(t0.x, t1.x), f_max(t0.y, t1.y), f_max(t0.z, t1.z));
    float tmin = f_max(f_max(f_max(lo.x, lo.y), lo.z), t_min);
    float tmax = f_min(f_min(f_min(hi.x, hi.y), hi.z), t_max);
    return tmax >= tmin;
}

/* kernel body :99-211; shadow != 0 is the -D SHADOW_RAYS variant */
static void TraceOne(orc* o, const rt_ray* rays, uint32_t ray_idx, int shadow, rt_hit* hit_out, uint32_t* shadow_out)
{
    rt_ray ray = rays[ray_idx];
    v3 org = V3(ray.origin.x, ray.origin.y, ray.origin.z);
    v3 dir = V3(ray.direction.x, ray.direction.y, ray.direction.z);
    float t_min = ray.origin.w;
    float t_max = ray.direction.w;
    v3 inv = V3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);               /* :125 */
    int ray_sign[3];
    ray_sign[0] = inv.x < 0; ray_sign[1] = inv.y < 0; ray_sign[2] = inv.z < 0;

    uint32_t shadow_hit = RT_INVALID_ID;
    rt_hit hit;
    memset(&hit, 0, sizeof(hit));   /* reference leaves bc/t uninitialised on a miss */
    hit.primitive_id = RT_INVALID_ID;

    int toVisitOffset = 0, currentNodeIndex = 0;
    int nodesToVisit[64];
    uint64_t n_nodes = 0, n_tris = 0;

    for (;;)
    {
        const rt_bvh_node* node = &o->nodes[currentNodeIndex];
        ++n_nodes;
        if (RayBounds(node, org, inv, t_min, t_max))
        {
            int num_primitives = (int)(node->num_primitives_axis >> 16);
            if (num_primitives > 0)
            {
                for (int i = 0; i < num_primitives; ++i)
                {
                    ++n_tris;
                    float u, v, t;
                    if (RayTriangle(org, dir, t_min, t_max, &o->triangles[node->offset + i], &u, &v, &t))
                    {
                        hit.bc.x = u; hit.bc.y = v; hit.t = t;
                        hit.primitive_id = node->offset + i;
                        t_max = t;                                       /* :162 */
                        if (shadow) { shadow_hit = 0; goto endtrace; }   /* :164-167 */
                    }
                }
                if (toVisitOffset == 0) break;
                currentNodeIndex = nodesToVisit[--toVisitOffset];
            }
            else
            {
                if (ray_sign[node->num_primitives_axis & 0xFFFF])        /* :181-190 */
                {
                    nodesToVisit[toVisitOffset++] = currentNodeIndex + 1;
                    currentNodeIndex = (int)node->offset;
                }
                else
                {
                    nodesToVisit[toVisitOffset++] = (int)node->offset;
                    currentNodeIndex = currentNodeIndex + 1;
                }
            }
        }
        else
        {
            if (toVisitOffset == 0) break;
            currentNodeIndex = nodesToVisit[--toVisitOffset];
        }
    }
endtrace:
    if (shadow)
    {
        *shadow_out = shadow_hit;
        o->shadow_nodes += n_nodes; o->shadow_tris += n_tris;
    }
    else
    {
        *hit_out = hit;
        o->closest_nodes += n_nodes; o->closest_tris += n_tris;
    }
}

/* ---- Miss, miss.cl:28-77 ------------------------------------------------ */
/* read_imagef with CLK_NORMALIZED_COORDS_TRUE | CLK_ADDRESS_REPEAT |
   CLK_FILTER_LINEAR (miss.cl:30): OpenCL 1.2 spec 8.2, evaluation order of
   cl_builtins.cpp shim_read_imagef */
static v3 ReadImageLinearRepeat(const orc* o, float cx, float cy)
{
    int w = (int)o->env_w, h = (int)o->env_h;
    float u = (cx - __builtin_floorf(cx)) * (float)w;
    float v = (cy - __builtin_floorf(cy)) * (float)h;
    float fu = __builtin_floorf(u - 0.5f);
    float fv = __builtin_floorf(v - 0.5f);
    /* OpenCL leaves read_imagef undefined for NaN coordinates (6.12.14); they DO occur: acos of a direction component one ulp
     * above 1 (miss.cl:34), once in ~1e9 escaped rays.  x86 converts NaN to INT_MIN (an address far outside the image: the
     * reference's kernels crashed here in a 128-spp run of the config-5 stand-in), gfx950 to 0.  Defined for the project: texel
     * (0, 0), and the NaN weights make the sample NaN -- what the GPU path always produced. */
    int i0 = fu != fu ? 0 : (int)fu, j0 = fv != fv ? 0 : (int)fv;
    int i1 = i0 + 1, j1 = j0 + 1;
    if (i0 < 0) i0 = w + i0;
    if (i1 > w - 1) i1 = i1 - w;
    if (j0 < 0) j0 = h + j0;
    if (j1 > h - 1) j1 = j1 - h;
    float a = (u - 0.5f) - fu;
    float b = (v - 0.5f) - fv;
    float wa0 = 1.0f - a, wb0 = 1.0f - b;
    const float* t00 = o->env + 4 * ((size_t)j0 * w + i0);
    const float* t10 = o->env + 4 * ((size_t)j0 * w + i1);
    const float* t01 = o->env + 4 * ((size_t)j1 * w + i0);
    const float* t11 = o->env + 4 * ((size_t)j1 * w + i1);
    float w00 = wa0 * wb0, w10 = a * wb0, w01 = wa0 * b, w11 = a * b;
    return V3(w00 * t00[0] + w10 * t10[0] + w01 * t01[0] + w11 * t11[0],
              w00 * t00[1] + w10 * t10[1] + w01 * t01[1] + w11 * t11[1],
              w00 * t00[2] + w10 * t10[2] + w01 * t01[2] + w11 * t11[2]);
}

/* SampleSky :28-39 */
static v3 SampleSky(const orc* o, v3 dir)
{
    float cx = rt_atan2f(dir.x, dir.y) + RT_PI;
    float cy = rt_acosf(dir.z);
    cx = cx < 0.0f ? cx + RT_TWO_PI : cx;
    cx *= RT_INV_TWO_PI;
    cy *= RT_INV_PI;
    return ReadImageLinearRepeat(o, cx, cy);
}

static void Miss(orc* o, uint32_t bounce, uint32_t ray_idx)
{
    uint32_t in = bounce & 1;
    if (ray_idx >= o->ray_counter[in]) return;
    rt_ray ray = o->rays[in][ray_idx];
    rt_hit hit = o->hits[ray_idx];
    if (hit.primitive_id == RT_INVALID_ID)
    {
        uint32_t pixel_idx = o->pixel_indices[in][ray_idx];
        v3 thr = f3(o->throughputs[pixel_idx]);
        v3 sky = o->furnace ? V3(0.5f, 0.5f, 0.5f)                        /* :70-71 */
                            : SampleSky(o, V3(ray.direction.x, ray.direction.y, ray.direction.z));
        v3 add = v_mul(sky, thr);
        o->radiance[pixel_idx].x += add.x;
        o->radiance[pixel_idx].y += add.y;
        o->radiance[pixel_idx].z += add.z;
        o->n_escaped++;
    }
}

/* ---- material library --------------------------------------------------- */
typedef struct
{
    v3 diffuse_albedo; float roughness;
    v3 specular_albedo; float metalness;
    v3 emission; float ior; float transparency;
} Material;                                                               /* material.h:35-49 */

/* utils.h:123-131, material.h:251-264 */
static v3 SampleTexture(orc* o, rt_texture tex, v2 uv)
{
    uv.x -= __builtin_floorf(uv.x);
    uv.y -= __builtin_floorf(uv.y);
    uv.y = 1.f - uv.y;
    int texel_x = i_clamp((int)(uv.x * (float)tex.width), 0, tex.width - 1);
    int texel_y = i_clamp((int)(uv.y * (float)tex.height), 0, tex.height - 1);
    int texel_addr = tex.data_start + texel_y * tex.width + texel_x;
    uint32_t data = o->texture_data[texel_addr];
    o->n_texels++;
    float r = (float)(data & 0xFF) / 255.0f;
    float g = (float)((data >> 8) & 0xFF) / 255.0f;
    float b = (float)((data >> 16) & 0xFF) / 255.0f;
    return V3(f_min(f_max(r, 0.0f), 1.0f), f_min(f_max(g, 0.0f), 1.0f), f_min(f_max(b, 0.0f), 1.0f));
}

static v3 pow3(v3 a, float e) { return V3(rt_powf(a.x, e), rt_powf(a.y, e), rt_powf(a.z, e)); }

/* utils.h:133-147 */
static v3 UnpackRGBTex(uint32_t data, uint32_t* idx)
{
    float r = (float)(data & 0xFF), g = (float)((data >> 8) & 0xFF), b = (float)((data >> 16) & 0xFF);
    *idx = (data >> 24) & 0xFF;
    return V3(r / 255.0f, g / 255.0f, b / 255.0f);
}

/* utils.h:149-158 */
static v3 UnpackRGBE(uint32_t rgbe)
{
    int r = (int)((rgbe >> 0) & 0xFF), g = (int)((rgbe >> 8) & 0xFF), b = (int)((rgbe >> 16) & 0xFF);
    int e = (int)(rgbe >> 24);
    float f = rt_ldexpf(1.0f, e - (128 + 8));
    return V3((float)r * f, (float)g * f, (float)b * f);
}

/* material.h:319-369; mtl = the material's index (wide texture indices, when the extension is on) */
static void ApplyTextures(orc* o, uint32_t mtl, Material* out, v2 uv)
{
    rt_packed_material in = o->materials[mtl];
    const uint16_t* wide = o->tex16 ? o->tex16 + (size_t)mtl * 6u : NULL;
    const uint32_t none = wide ? 0xFFFFu : RT_INVALID_TEXTURE_IDX;
    uint32_t idx;
    out->diffuse_albedo = UnpackRGBTex(in.diffuse_albedo, &idx);
    if (wide) idx = wide[0];
    if (idx != none) out->diffuse_albedo = pow3(SampleTexture(o, o->textures[idx], uv), 2.2f);
    out->specular_albedo = UnpackRGBTex(in.specular_albedo, &idx);
    if (wide) idx = wide[1];
    if (idx != none) out->specular_albedo = pow3(SampleTexture(o, o->textures[idx], uv), 2.2f);
    out->emission = UnpackRGBE(in.emission);

    uint32_t d = in.roughness_metalness;                                  /* utils.h:160-174 */
    out->roughness = (float)((d >> 0) & 0xFF) / 255.0f;
    uint32_t roughness_idx = wide ? wide[2] : (d >> 8) & 0xFF;
    out->metalness = (float)((d >> 16) & 0xFF) / 255.0f;
    uint32_t metalness_idx = wide ? wide[3] : (d >> 24) & 0xFF;
    if (roughness_idx != none) out->roughness = SampleTexture(o, o->textures[roughness_idx], uv).x;
    if (metalness_idx != none) out->metalness = SampleTexture(o, o->textures[metalness_idx], uv).x;

    d = in.ior_emission_idx_transparency;                                 /* utils.h:176-190 */
    out->ior = (float)((d >> 0) & 0xFF) / 25.5f;
    uint32_t emission_idx = wide ? wide[4] : (d >> 8) & 0xFF;
    out->transparency = (float)((d >> 16) & 0xFF) / 255.0f;
    uint32_t transparency_idx = wide ? wide[5] : (d >> 24) & 0xFF;
    if (emission_idx != none)
        out->emission = v_mul(out->emission, pow3(SampleTexture(o, o->textures[emission_idx], uv), 2.2f));
    if (transparency_idx != none)
        out->transparency *= SampleTexture(o, o->textures[transparency_idx], uv).x;
}

/* bxdf.h:57-61 */
static float IorToF0(float ior_incident, float ior_transmitted)
{
    float result = (ior_transmitted - ior_incident) / (ior_transmitted + ior_incident);
    return result * result;
}

/* bxdf.h:71-74 */
static v3 FresnelSchlick(v3 f0, float h_dot_o)
{
    float p = rt_powf(1.0f - h_dot_o, 5.0f);
    return V3(f0.x + (1.0f - f0.x) * p, f0.y + (1.0f - f0.y) * p, f0.z + (1.0f - f0.z) * p);
}

/* bxdf.h:90-95 */
static float GGX_D(float alpha, float n_dot_h)
{
    float alpha2 = alpha * alpha;
    float denom = n_dot_h * n_dot_h * (alpha2 - 1.0f) + 1.0f;
    return alpha2 * RT_INV_PI / (denom * denom);
}

/* bxdf.h:104-119 */
static float V_SmithGGXCorrelated(float n_dot_i, float n_dot_o, float alphaG)
{
    float alphaG2 = alphaG * alphaG;
    float Lambda_GGXV = n_dot_o * __builtin_sqrtf((-n_dot_i * alphaG2 + n_dot_i) * n_dot_i + alphaG2);
    float Lambda_GGXL = n_dot_i * __builtin_sqrtf((-n_dot_o * alphaG2 + n_dot_o) * n_dot_o + alphaG2);
    return 0.5f / (Lambda_GGXV + Lambda_GGXL);
}

static float Luma(v3 rgb) { return rgb.x * 0.299f + rgb.y * 0.587f + rgb.z * 0.114f; }   /* utils.h:108-111 */

/* utils.h:99-106 */
static v3 TangentToWorld(v3 dir, v3 n)
{
    v3 axis = __builtin_fabsf(n.x) > 0.001f ? V3(0.0f, 1.0f, 0.0f) : V3(1.0f, 0.0f, 0.0f);
    v3 t = v_normalize(v_cross(axis, n));
    v3 b = v_cross(n, t);
    return v_normalize(v_add(v_add(v_scale(b, dir.x), v_scale(t, dir.y)), v_scale(n, dir.z)));
}

static v3 reflect(v3 v, v3 n) { return v_sub(v, v_scale(n, 2.0f * v_dot(v, n))); }      /* utils.h:83-86 */

/* bxdf.h:157-168 */
static v3 GGX_Sample(v2 s, v3 n, float alpha)
{
    float phi = RT_TWO_PI * s.x;
    /* double-precision literals in the reference: 1.0 + (float)/(1.0 - s.y) -> fp64 sqrt, fp64 divide */
    float cos_theta = (float)(1.0 / __builtin_sqrt(1.0 + (double)(alpha * alpha * s.y) / (1.0 - (double)s.y)));
    float sin_theta = __builtin_sqrtf(f_max(0.0f, 1.0f - cos_theta * cos_theta));
    v3 axis = __builtin_fabsf(n.x) > 0.001f ? V3(0.0f, 1.0f, 0.0f) : V3(1.0f, 0.0f, 0.0f);
    v3 t = v_normalize(v_cross(axis, n));
    v3 b = v_cross(n, t);
    float cp = rt_cosf(phi), sp = rt_sinf(phi);
    v3 r = v_add(v_add(v_scale(v_scale(b, cp), sin_theta), v_scale(v_scale(t, sp), sin_theta)), v_scale(n, cos_theta));
    return v_normalize(r);
}

/* material.h:132-169 */
static v3 EvaluateMaterial(const Material* m, v3 normal, v3 incoming, v3 outgoing)
{
    if ((double)m->transparency < 0.5) return V3(0.0f, 0.0f, 0.0f);
    v3 half_vec = v_normalize(v_add(incoming, outgoing));
    float n_dot_i = f_max(v_dot(normal, incoming), RT_EPS);
    float n_dot_o = f_max(v_dot(normal, outgoing), RT_EPS);
    float n_dot_h = f_max(v_dot(normal, half_vec), RT_EPS);
    float h_dot_o = f_max(v_dot(half_vec, outgoing), RT_EPS);
    float alpha = m->roughness * m->roughness;
    float f0_dielectric = IorToF0(1.0f, m->ior);
    v3 f0 = v_mix(V3(f0_dielectric, f0_dielectric, f0_dielectric), m->specular_albedo, m->metalness);
    v3 diffuse_color = v_scale(m->diffuse_albedo, 1.0f - m->metalness);
    v3 fresnel = FresnelSchlick(f0, h_dot_o);
    float specular = GGX_D(alpha, n_dot_h) * V_SmithGGXCorrelated(n_dot_i, n_dot_o, alpha);   /* :119-125 */
    v3 diffuse = v_scale(diffuse_color, RT_INV_PI);
    return V3(fresnel.x * specular + (1.0f - fresnel.x) * diffuse.x,
              fresnel.y * specular + (1.0f - fresnel.y) * diffuse.y,
              fresnel.z * specular + (1.0f - fresnel.z) * diffuse.z);
}

/* material.h:171-241 (with SampleSpecular :66-103, SampleDiffuse :51-64, SampleTransparency :105-117) */
static v3 SampleBxdf(const orc* o, float s1, v2 s, Material material, v3 normal, v3 incoming, v3* outgoing,
    float* pdf, float* offset, int* delta /* extension: the event chosen has a delta distribution */)
{
    *delta = 0;
    if (o->furnace)
    {
        material.diffuse_albedo = V3(1.0f, 1.0f, 1.0f);
        material.specular_albedo = V3(1.0f, 1.0f, 1.0f);
    }
    float alpha = material.roughness * material.roughness;
    float f0_dielectric = IorToF0(1.0f, material.ior);
    v3 f0 = v_mix(V3(f0_dielectric, f0_dielectric, f0_dielectric), material.specular_albedo, material.metalness);
    v3 diffuse_albedo = v_scale(material.diffuse_albedo, 1.0f - material.metalness);
    v3 specular_albedo = v_mix(material.specular_albedo, V3(1.0f, 1.0f, 1.0f), material.metalness);
    v3 fresnel = v_mul(FresnelSchlick(f0, v_dot(normal, incoming)), specular_albedo);
    float specular_weight = Luma(v_mul(specular_albedo, fresnel));
    float diffuse_weight = Luma(v_mul(diffuse_albedo, V3(1.0f - fresnel.x, 1.0f - fresnel.y, 1.0f - fresnel.z)));
    float weight_sum = diffuse_weight + specular_weight;
    float specular_sampling_pdf = specular_weight / weight_sum;
    float diffuse_sampling_pdf = diffuse_weight / weight_sum;

    *offset = 1.0f;
    if ((double)material.transparency < 0.5)
    {
        *pdf = 1.0f;
        *outgoing = v_neg(incoming);
        *offset = -1.0f;
        *delta = 1;
        return V3(1.0f, 1.0f, 1.0f);
    }

    v3 bxdf;
    if (s1 <= specular_sampling_pdf)
    {
        float spec;
        if (alpha <= 1e-4f)
        {
            *outgoing = reflect(v_neg(incoming), normal);
            *pdf = 1.0f;
            float n_dot_o = v_dot(*outgoing, normal);
            spec = 1.0f / n_dot_o;
            *delta = 1;
        }
        else
        {
            v3 wh = GGX_Sample(s, normal, alpha);
            *outgoing = reflect(v_neg(incoming), wh);
            float n_dot_o = v_dot(normal, *outgoing);
            float n_dot_h = v_dot(normal, wh);
            float n_dot_i = v_dot(normal, incoming);
            float D = GGX_D(alpha, n_dot_h);
            float G = V_SmithGGXCorrelated(n_dot_i, n_dot_o, alpha);
            *pdf = D * n_dot_h / (4.0f * v_dot(wh, *outgoing));
            spec = D * G;
        }
        float m = f_max(v_dot(*outgoing, normal), 0.0f);
        bxdf = V3(fresnel.x * spec * m, fresnel.y * spec * m, fresnel.z * spec * m);
        *pdf *= specular_sampling_pdf;
    }
    else
    {
        /* SampleHemisphereCosine bxdf.h:33-54 */
        float phi = RT_TWO_PI * s.x;
        float sin_theta = __builtin_sqrtf(s.y);
        float cos_theta = __builtin_sqrtf(1.0f - s.y);
        *pdf = cos_theta * RT_INV_PI;
        v3 tbn = V3(rt_cosf(phi) * sin_theta, rt_sinf(phi) * sin_theta, cos_theta);
        *outgoing = TangentToWorld(tbn, normal);
        v3 d = v_scale(diffuse_albedo, RT_INV_PI);
        float m = f_max(v_dot(*outgoing, normal), 0.0f);
        bxdf = V3((1.0f - fresnel.x) * d.x * m, (1.0f - fresnel.y) * d.y * m, (1.0f - fresnel.z) * d.z * m);
        *pdf *= diffuse_sampling_pdf;
    }
    return bxdf;
}

/* light.h:30-65 */
static v3 Light_Sample(const orc* o, v3 position, float s, v3* outgoing, float* pdf)
{
    uint32_t count = o->scene_info.analytic_light_count;
    int light_idx = i_clamp((int)(s * (float)count), 0, (int)count - 1);
    rt_light light = o->lights[light_idx];
    *pdf = 1.0f / (float)count;
    v3 light_radiance = f3(light.radiance);
    if (light.type == RT_LIGHT_TYPE_POINT)
    {
        v3 to_light = v_sub(f3(light.origin), position);
        float sq_length = v_dot(to_light, to_light);
        light_radiance = v_divs(light_radiance, sq_length);
        *outgoing = to_light;
    }
    else
    {
        *outgoing = v_scale(f3(light.origin), RT_MAX_RENDER_DIST);
    }
    return light_radiance;
}

/* Extension RT_SCENE_EMISSIVE_NEE (not in the reference: scene.cpp:324-339 collects the emissive triangles,
 * hit_surface.cl:39 receives and ignores them).  Light_Sample over n = analytic lights + emissive triangles: index
 * from s as in light.h:41, selection pdf 1 / n.  An emissive triangle is sampled uniformly by area with u1 = the
 * fraction of s * n left over by the index and u2 = the BSDF-layer sample of this bounce; it emits from its front side
 * only, as in the reference (whose traversal culls back faces).  Returned: radiance * cos_l * area / d^2, so
 * that the caller's  radiance * throughput * brdf / pdf * cos  is the area-sampling estimator; `outgoing` stops short
 * of the sampled point, so that the shadow ray does not see the emitter itself (below). */
static v3 Light_SampleWithEmissive(orc* o, v3 position, float s, float s2, v3* outgoing, float* pdf)
{
    const uint32_t n_analytic = o->scene_info.analytic_light_count;
    const uint32_t count = n_analytic + o->n_emissive_tris;
    int light_idx = i_clamp((int)(s * (float)count), 0, (int)count - 1);
    *pdf = 1.0f / (float)count;
    if ((uint32_t)light_idx < n_analytic)
    {
        rt_light light = o->lights[light_idx];
        v3 light_radiance = f3(light.radiance);
        if (light.type == RT_LIGHT_TYPE_POINT)
        {
            v3 to_light = v_sub(f3(light.origin), position);
            float sq_length = v_dot(to_light, to_light);
            light_radiance = v_divs(light_radiance, sq_length);
            *outgoing = to_light;
        }
        else
            *outgoing = v_scale(f3(light.origin), RT_MAX_RENDER_DIST);
        return light_radiance;
    }
    const rt_triangle* tri = &o->triangles[o->emissive[(uint32_t)light_idx - n_analytic]];
    float u1 = s * (float)count - (float)light_idx;
    u1 = f_min(f_max(u1, 0.0f), 1.0f);
    float su = __builtin_sqrtf(u1);
    float b0 = 1.0f - su, b1 = s2 * su;
    float b2 = 1.0f - b0 - b1;
    v3 p1 = f3(tri->v1.position), p2 = f3(tri->v2.position), p3 = f3(tri->v3.position);
    v3 lp = v_add(v_add(v_scale(p1, b0), v_scale(p2, b1)), v_scale(p3, b2));
    v2 uv;
    uv.x = tri->v1.texcoord.x * b0 + tri->v2.texcoord.x * b1 + tri->v3.texcoord.x * b2;
    uv.y = tri->v1.texcoord.y * b0 + tri->v2.texcoord.y * b1 + tri->v3.texcoord.y * b2;
    Material lm;
    ApplyTextures(o, tri->mtl_index, &lm, uv);
    v3 nl = v_cross(v_sub(p2, p1), v_sub(p3, p1));                        /* length = 2 * area */
    v3 to_light = v_sub(lp, position);
    float d2 = v_dot(to_light, to_light);
    float g = 0.0f;
    /* The reference's ray-triangle test culls back faces (trace_bvh.cl:28-73: det = -dir . nl < 1e-8 -> no hit), so a
     * triangle is visible -- and its emission counted (hit_surface.cl:107-112) -- only from the side its normal
     * points to.  The same rule here: */
    const float dist = __builtin_sqrtf(d2);
    const float nd = -v_dot(nl, to_light);                                /* 2 * area * d * cos_l, > 0 on the front side */
    if (d2 > 0.0f && nd / dist >= 1e-8f) g = (nd * 0.5f) / (dist * d2);   /* cos_l * area / d^2 */
    /* The shadow ray starts at position + normal * EPS (hit_surface.cl:131) but is aimed from `position`: it reaches the
     * emitter's plane up to EPS / |cos_l| EARLIER than d.  Stop short by twice that plus 2^-10 d, as a fraction of d;
     * samples for which nothing is left (grazing or touching the emitter: |cos_l| * area / d^2 -> 0 there) are dropped. */
    float keep = 1.0f - 0.0009765625f - (2.0f * RT_EPS) * __builtin_sqrtf(v_dot(nl, nl)) / nd;
    if (!(keep > 0.0f) || !(g > 0.0f)) { keep = 1.0f; g = 0.0f; }
    *outgoing = v_scale(to_light, keep);
    return v_scale(lm.emission, g);
}

/* ---- HitSurface, hit_surface.cl:30-186 ---------------------------------- */
static void HitSurface(orc* o, uint32_t bounce, uint32_t ray_idx)
{
    uint32_t in = bounce & 1, out = (bounce + 1) & 1;
    if (ray_idx >= o->ray_counter[in]) return;
    rt_hit hit = o->hits[ray_idx];
    if (hit.primitive_id == RT_INVALID_ID) return;
    o->n_hits++;

    rt_ray incoming_ray = o->rays[in][ray_idx];
    v3 incoming = V3(-incoming_ray.direction.x, -incoming_ray.direction.y, -incoming_ray.direction.z);
    uint32_t pixel_idx = o->pixel_indices[in][ray_idx];
    uint32_t sample_idx = o->sample_counter;
    int x = (int)(pixel_idx % o->width);
    int y = (int)(pixel_idx / o->width);

    const rt_triangle* tri = &o->triangles[hit.primitive_id];
    float bu = hit.bc.x, bv = hit.bc.y;
    float w0 = 1.0f - bu - bv;
    v3 p1 = f3(tri->v1.position), p2 = f3(tri->v2.position), p3 = f3(tri->v3.position);
    v3 position = v_add(v_add(v_scale(p1, w0), v_scale(p2, bu)), v_scale(p3, bv));     /* utils.h:94-97 */
    v3 geometry_normal = v_normalize(v_cross(v_sub(p2, p1), v_sub(p3, p1)));
    v2 texcoord;
    texcoord.x = tri->v1.texcoord.x * w0 + tri->v2.texcoord.x * bu + tri->v3.texcoord.x * bv;
    texcoord.y = tri->v1.texcoord.y * w0 + tri->v2.texcoord.y * bu + tri->v3.texcoord.y * bv;
    v3 normal = v_normalize(v_add(v_add(v_scale(f3(tri->v1.normal), w0), v_scale(f3(tri->v2.normal), bu)),
        v_scale(f3(tri->v3.normal), bv)));

    Material material;
    ApplyTextures(o, tri->mtl_index, &material, texcoord);
    v3 hit_throughput = f3(o->throughputs[pixel_idx]);

    /* Extension RT_SCENE_EMISSIVE_NEE: light that next-event estimation already gathers from the emissive triangles must
     * not be counted again when a scattered ray happens to hit one: emission is added for camera rays and after delta
     * events (which next-event estimation cannot sample) only. */
    const int count_emission = !o->emissive_nee || bounce == 0 || o->prev_delta[pixel_idx];
    if (!o->furnace && count_emission)                                    /* :107-112 */
    {
        if (material.emission.x * 1.0f + material.emission.y * 1.0f + material.emission.z * 1.0f > 0.0f)
        {
            v3 e = v_mul(hit_throughput, material.emission);
            o->radiance[pixel_idx].x += e.x;
            o->radiance[pixel_idx].y += e.y;
            o->radiance[pixel_idx].z += e.z;
            o->n_emissive++;
        }
    }

    /* Direct lighting :115-145 */
    {
        float s_light = SampleRandom((uint32_t)x, (uint32_t)y, sample_idx, bounce, 4);
        v3 outgoing;
        float pdf;
        v3 light_radiance;
        if (!o->emissive_nee)
            light_radiance = Light_Sample(o, position, s_light, &outgoing, &pdf);
        else
        {
            /* one light out of the analytic lights AND the emissive triangles, uniformly */
            float s_layer = SampleRandom((uint32_t)x, (uint32_t)y, sample_idx, bounce, 1);
            light_radiance = Light_SampleWithEmissive(o, position, s_light, s_layer, &outgoing, &pdf);
        }
        float distance_to_light = v_length(outgoing);
        outgoing = v_normalize(outgoing);
        v3 brdf = EvaluateMaterial(&material, normal, incoming, outgoing);
        float m = f_max(v_dot(outgoing, normal), 0.0f);
        v3 ls = v_scale(v_divs(v_mul(v_mul(light_radiance, hit_throughput), brdf), pdf), m);
        int spawn_shadow_ray = (pdf > 0.0f) && (v_dot(ls, ls) > 0.0f);
        if (spawn_shadow_ray)
        {
            uint32_t idx = o->shadow_ray_counter++;
            rt_ray sr;
            v3 so = v_add(position, v_scale(normal, RT_EPS));
            sr.origin.x = so.x; sr.origin.y = so.y; sr.origin.z = so.z; sr.origin.w = 0.0f;
            sr.direction.x = outgoing.x; sr.direction.y = outgoing.y; sr.direction.z = outgoing.z;
            sr.direction.w = distance_to_light;
            o->shadow_rays[idx] = sr;
            o->shadow_pixel_indices[idx] = pixel_idx;
            o->direct_light_samples[idx].x = ls.x;
            o->direct_light_samples[idx].y = ls.y;
            o->direct_light_samples[idx].z = ls.z;
        }
    }

    /* Indirect lighting :148-184 */
    {
        v2 s;
        s.x = SampleRandom((uint32_t)x, (uint32_t)y, sample_idx, bounce, 2);
        s.y = SampleRandom((uint32_t)x, (uint32_t)y, sample_idx, bounce, 3);
        float s1 = SampleRandom((uint32_t)x, (uint32_t)y, sample_idx, bounce, 1);
        float pdf = 0.0f;
        v3 throughput = V3(0.0f, 0.0f, 0.0f);
        v3 outgoing;
        float offset;
        int delta;
        v3 bxdf = SampleBxdf(o, s1, s, material, normal, incoming, &outgoing, &pdf, &offset, &delta);
        if (o->prev_delta) o->prev_delta[pixel_idx] = (uint8_t)delta;
        if ((double)pdf > 0.0) throughput = v_divs(bxdf, pdf);
        o->throughputs[pixel_idx].x *= throughput.x;
        o->throughputs[pixel_idx].y *= throughput.y;
        o->throughputs[pixel_idx].z *= throughput.z;
        if ((double)pdf > 0.0)
        {
            uint32_t idx = o->ray_counter[out]++;
            v3 oo = v_add(position, v_scale(v_scale(geometry_normal, RT_EPS), offset));
            rt_ray r;
            r.origin.x = oo.x; r.origin.y = oo.y; r.origin.z = oo.z; r.origin.w = 0.0f;
            r.direction.x = outgoing.x; r.direction.y = outgoing.y; r.direction.z = outgoing.z;
            r.direction.w = RT_MAX_RENDER_DIST;
            o->rays[out][idx] = r;
            o->pixel_indices[out][idx] = pixel_idx;
            o->n_outgoing++;
        }
    }
}

/* ---- GenerateAOV, aov.cl:30-110 ----------------------------------------- */
static v2 ProjectScreen(v3 position, const rt_camera* cam)              /* aov.cl:30-42 */
{
    v3 d = v_normalize(v_sub(position, f3(cam->position)));
    v3 ipd = v_divs(d, v_dot(f3(cam->front), d));
    float angle = rt_tanf(0.5f * cam->fov);
    v3 right = v_cross(f3(cam->front), f3(cam->up));
    float u = v_dot(right, ipd) / (angle * cam->aspect_ratio);
    float v = v_dot(f3(cam->up), ipd) / (angle);
    v2 r;
    r.x = u * 0.5f + 0.5f;
    r.y = v * 0.5f + 0.5f;
    return r;
}

static void GenerateAOV(orc* o, uint32_t ray_idx)
{
    if (ray_idx >= o->ray_counter[0]) return;
    rt_hit hit = o->hits[ray_idx];
    if (hit.primitive_id == RT_INVALID_ID) return;
    rt_ray ray = o->rays[0][ray_idx];
    uint32_t pixel_idx = o->pixel_indices[0][ray_idx];
    const rt_triangle* tri = &o->triangles[hit.primitive_id];
    float bu = hit.bc.x, bv = hit.bc.y;
    float w0 = 1.0f - bu - bv;
    v3 p1 = f3(tri->v1.position), p2 = f3(tri->v2.position), p3 = f3(tri->v3.position);
    v3 position = v_add(v_add(v_scale(p1, w0), v_scale(p2, bu)), v_scale(p3, bv));
    v2 texcoord;
    texcoord.x = tri->v1.texcoord.x * w0 + tri->v2.texcoord.x * bu + tri->v3.texcoord.x * bv;
    texcoord.y = tri->v1.texcoord.y * w0 + tri->v2.texcoord.y * bu + tri->v3.texcoord.y * bv;
    v3 normal = v_normalize(v_add(v_add(v_scale(f3(tri->v1.normal), w0), v_scale(f3(tri->v2.normal), bu)),
        v_scale(f3(tri->v3.normal), bv)));
    Material material;
    ApplyTextures(o, tri->mtl_index, &material, texcoord);
    o->diffuse_albedo[pixel_idx].x = material.diffuse_albedo.x;
    o->diffuse_albedo[pixel_idx].y = material.diffuse_albedo.y;
    o->diffuse_albedo[pixel_idx].z = material.diffuse_albedo.z;
    o->depth[pixel_idx] = v_length(v_sub(V3(ray.origin.x, ray.origin.y, ray.origin.z), position));
    o->normal[pixel_idx].x = normal.x; o->normal[pixel_idx].y = normal.y; o->normal[pixel_idx].z = normal.z;
    v2 a = ProjectScreen(position, &o->camera), b = ProjectScreen(position, &o->aov_prev_camera);
    o->velocity[pixel_idx].x = a.x - b.x;
    o->velocity[pixel_idx].y = a.y - b.y;
}

/* ---- TemporalAccumulation, denoiser.cl:27-79 ---------------------------- */
static void TemporalAccumulation(orc* o, uint32_t pixel_idx)
{
    uint32_t width = o->width, height = o->height;
    int x = (int)(pixel_idx % width);
    int y = (int)(pixel_idx / width);
    if ((uint32_t)x >= width || (uint32_t)y >= height) return;
    float depth_value = o->depth[pixel_idx];
    if (depth_value == RT_MAX_RENDER_DIST) return;
    float mx = o->velocity[pixel_idx].x, my = o->velocity[pixel_idx].y;
    float prev_u = ((float)x + 0.5f) / (float)width - mx;
    float prev_v = ((float)y + 0.5f) / (float)height - my;
    int prev_x = (int)(prev_u * (float)width);
    int prev_y = (int)(prev_v * (float)height);
    /* the comparisons against the unsigned width/height promote prev_x to unsigned (C rules), so a
       negative prev_x fails the `>= width` test as well -- same outcome as written */
    if (prev_x < 0 || (uint32_t)prev_x >= width || prev_y < 0 || (uint32_t)prev_y >= height) return;
    int prev_idx = prev_y * (int)width + prev_x;
    float prev_depth_value = o->prev_depth[prev_idx];
    if (__builtin_fabsf(depth_value - prev_depth_value) / depth_value > 0.1f) return;
    v3 cur = V3(o->radiance[pixel_idx].x, o->radiance[pixel_idx].y, o->radiance[pixel_idx].z);
    v3 prev = V3(o->prev_radiance[prev_idx].x, o->prev_radiance[prev_idx].y, o->prev_radiance[prev_idx].z);
    v3 m = v_mix(cur, prev, 0.9f);
    o->radiance[pixel_idx].x = m.x; o->radiance[pixel_idx].y = m.y; o->radiance[pixel_idx].z = m.z;
}

/* ---- AccumulateDirectSamples, accumulate_direct_samples.cl:27-53 -------- */
static void AccumulateDirectSamples(orc* o, uint32_t ray_idx)
{
    if (ray_idx >= o->shadow_ray_counter) return;
    if (o->shadow_hits[ray_idx] == RT_INVALID_ID)
    {
        uint32_t pixel_idx = o->shadow_pixel_indices[ray_idx];
        o->radiance[pixel_idx].x += o->direct_light_samples[ray_idx].x;
        o->radiance[pixel_idx].y += o->direct_light_samples[ray_idx].y;
        o->radiance[pixel_idx].z += o->direct_light_samples[ray_idx].z;
        o->n_unoccluded++;
    }
}

/* ---- host schedule ------------------------------------------------------ */
#define ORC_EXPORT __attribute__((visibility("default")))

ORC_EXPORT void* orc_create(uint32_t width, uint32_t height, int white_furnace)
{
    orc* o = (orc*)calloc(1, sizeof(orc));
    size_t n = (size_t)width * height;
    o->width = width; o->height = height; o->furnace = white_furnace;
    o->max_bounces = 3;                                                   /* integrator.hpp:91 */
    o->radiance = (rt_float4*)calloc(n, sizeof(rt_float4));
    for (int i = 0; i < 2; ++i)
    {
        o->rays[i] = (rt_ray*)calloc(n, sizeof(rt_ray));
        o->pixel_indices[i] = (uint32_t*)calloc(n, 4);
    }
    o->shadow_rays = (rt_ray*)calloc(n, sizeof(rt_ray));
    o->shadow_pixel_indices = (uint32_t*)calloc(n, 4);
    o->hits = (rt_hit*)calloc(n, sizeof(rt_hit));
    o->shadow_hits = (uint32_t*)calloc(n, 4);
    o->throughputs = (rt_float3*)calloc(n, sizeof(rt_float3));
    o->direct_light_samples = (rt_float4*)calloc(n, sizeof(rt_float4));
    o->resolved = (rt_float4*)calloc(n, sizeof(rt_float4));
    o->diffuse_albedo = (rt_float3*)calloc(n, sizeof(rt_float3));
    o->depth = (float*)calloc(n, sizeof(float));
    o->normal = (rt_float3*)calloc(n, sizeof(rt_float3));
    o->velocity = (rt_float2*)calloc(n, sizeof(rt_float2));
    o->prev_radiance = (rt_float4*)calloc(n, sizeof(rt_float4));
    o->prev_depth = (float*)calloc(n, sizeof(float));
    return o;
}

ORC_EXPORT void orc_destroy(void* h)
{
    orc* o = (orc*)h;
    free(o->radiance);
    for (int i = 0; i < 2; ++i) { free(o->rays[i]); free(o->pixel_indices[i]); }
    free(o->shadow_rays); free(o->shadow_pixel_indices); free(o->hits); free(o->shadow_hits);
    free(o->throughputs); free(o->direct_light_samples); free(o->resolved);
    free(o->diffuse_albedo); free(o->depth); free(o->normal); free(o->velocity); free(o->prev_radiance); free(o->prev_depth);
    free(o->triangles); free(o->nodes); free(o->materials); free(o->textures); free(o->texture_data);
    free(o->lights); free(o->env);
    free(o->tex16); free(o->emissive); free(o->prev_delta);
    free(o);
}

static void* dup_mem(const void* p, size_t bytes)
{
    void* q = malloc(bytes ? bytes : 1);
    if (bytes) memcpy(q, p, bytes);
    return q;
}

/* CLPathTraceIntegrator::UploadGPUData, cl_pt_integrator.cpp:373-456 */
ORC_EXPORT void orc_upload(void* h, const rt_triangle* tris, uint32_t ntris, const rt_bvh_node* nodes, uint32_t nnodes,
    const rt_packed_material* mats, uint32_t nmats, const rt_texture* tex, uint32_t ntex,
    const uint32_t* texdata, uint32_t ntexdata, const rt_light* lights, uint32_t nlights,
    const uint32_t* emissive, uint32_t nemissive, const float* env_rgba, uint32_t env_w, uint32_t env_h)
{
    orc* o = (orc*)h;
    /* emissive_indices is bound but never read by the reference's kernels (hit_surface.cl:39); kept for the extension */
    o->emissive = (uint32_t*)dup_mem(emissive, (size_t)nemissive * 4); o->n_emissive_tris = nemissive;
    o->triangles = (rt_triangle*)dup_mem(tris, (size_t)ntris * sizeof(rt_triangle)); o->n_triangles = ntris;
    o->nodes = (rt_bvh_node*)dup_mem(nodes, (size_t)nnodes * sizeof(rt_bvh_node)); o->n_nodes = nnodes;
    o->materials = (rt_packed_material*)dup_mem(mats, (size_t)nmats * sizeof(rt_packed_material)); o->n_materials = nmats;
    o->textures = (rt_texture*)dup_mem(tex, (size_t)ntex * sizeof(rt_texture)); o->n_textures = ntex;
    o->texture_data = (uint32_t*)dup_mem(texdata, (size_t)ntexdata * 4); o->n_texture_data = ntexdata;
    o->lights = (rt_light*)dup_mem(lights, (size_t)nlights * sizeof(rt_light)); o->n_lights = nlights;
    o->env = (float*)dup_mem(env_rgba, (size_t)env_w * env_h * 16); o->env_w = env_w; o->env_h = env_h;
    o->scene_info.analytic_light_count = nlights;
    o->scene_info.emissive_count = nemissive;
}

/* opt-in extensions (rt_scene_desc::material_texture_indices / flags); call after orc_upload */
ORC_EXPORT void orc_set_extensions(void* h, const uint16_t* material_texture_indices, uint32_t flags)
{
    orc* o = (orc*)h;
    free(o->tex16); o->tex16 = NULL;
    if (material_texture_indices)
        o->tex16 = (uint16_t*)dup_mem(material_texture_indices, (size_t)o->n_materials * 6 * sizeof(uint16_t));
    o->emissive_nee = (flags & 1u) && o->n_emissive_tris ? 1 : 0;
    free(o->prev_delta); o->prev_delta = NULL;
    if (o->emissive_nee) o->prev_delta = (uint8_t*)calloc((size_t)o->width * o->height, 1);
}

ORC_EXPORT void orc_set_camera(void* h, const rt_camera* cam)           /* cl_pt_integrator.cpp:365-371 */
{
    orc* o = (orc*)h;
    o->camera = *cam;
    o->aov_prev_camera = o->prev_camera;
    o->prev_camera = *cam;
}
ORC_EXPORT void orc_enable_denoiser(void* h, int enable)                /* :485-495 */
{
    orc* o = (orc*)h;
    if ((enable != 0) == (o->denoiser != 0)) return;
    o->denoiser = enable != 0;
    o->request_reset = 1;
}
/* Integrator::SetSamplerType (cl_pt_integrator.cpp:458-468).  Process-wide in this oracle. */
static int* g_bn_storage;
ORC_EXPORT void orc_set_blue_noise_tables(const int* sobol, const int* scrambling, const int* ranking)
{
    free(g_bn_storage);
    g_bn_storage = (int*)calloc(65536 + 131072 + 131072 + 256, sizeof(int));
    memcpy(g_bn_storage, sobol, 65536 * sizeof(int));
    memcpy(g_bn_storage + 65536, scrambling, 131072 * sizeof(int));
    memcpy(g_bn_storage + 65536 + 131072, ranking, 131072 * sizeof(int));
    g_bn_sobol = g_bn_storage; g_bn_scramble = g_bn_storage + 65536; g_bn_rank = g_bn_storage + 65536 + 131072;
}
ORC_EXPORT void orc_set_sampler(void* h, int blue_noise)
{
    orc* o = (orc*)h;
    if ((blue_noise != 0) == (o->blue_noise != 0)) return;
    o->blue_noise = blue_noise != 0;
    o->request_reset = 1;
}
ORC_EXPORT void orc_set_aov(void* h, uint32_t aov)                      /* :470-483 */
{
    orc* o = (orc*)h;
    if (aov == o->aov) return;
    o->aov = aov;
    o->request_reset = 1;
}
ORC_EXPORT void orc_set_max_bounces(void* h, uint32_t b) { ((orc*)h)->max_bounces = b; ((orc*)h)->request_reset = 1; }
ORC_EXPORT void orc_request_reset(void* h) { ((orc*)h)->request_reset = 1; }

/* stages == the protected virtuals of Integrator (integrator.hpp:65-79) */
ORC_EXPORT void orc_stage_reset(void* h)                                  /* cl_pt_integrator.cpp:497-508 */
{
    orc* o = (orc*)h;
    if (!o->denoiser) o->sample_counter = 0;
    memset(o->radiance, 0, (size_t)o->width * o->height * sizeof(rt_float4));
}
ORC_EXPORT void orc_stage_generate_rays(void* h)
{
    orc* o = (orc*)h;
    uint32_t n = o->width * o->height;
    for (uint32_t i = 0; i < n; ++i) RayGeneration(o, i);
}
ORC_EXPORT void orc_stage_intersect(void* h, uint32_t bounce)
{
    orc* o = (orc*)h;
    uint32_t in = bounce & 1, n = o->ray_counter[in];
    for (uint32_t i = 0; i < n; ++i) TraceOne(o, o->rays[in], i, 0, &o->hits[i], NULL);
}
ORC_EXPORT void orc_stage_shade_miss(void* h, uint32_t bounce)
{
    orc* o = (orc*)h;
    uint32_t n = o->width * o->height;
    for (uint32_t i = 0; i < n; ++i) Miss(o, bounce, i);
}
ORC_EXPORT void orc_stage_clear_counters(void* h, uint32_t bounce)
{
    orc* o = (orc*)h;
    o->ray_counter[(bounce + 1) & 1] = 0;
    o->shadow_ray_counter = 0;
}
ORC_EXPORT void orc_stage_shade_hits(void* h, uint32_t bounce)
{
    orc* o = (orc*)h;
    g_blue_noise = o->blue_noise;   /* kernel variant -D BLUE_NOISE_SAMPLER (cl_pt_integrator.cpp:273-276) */
    uint32_t n = o->width * o->height;
    for (uint32_t i = 0; i < n; ++i) HitSurface(o, bounce, i);
}
ORC_EXPORT void orc_stage_intersect_shadow(void* h)
{
    orc* o = (orc*)h;
    uint32_t n = o->shadow_ray_counter;
    for (uint32_t i = 0; i < n; ++i) TraceOne(o, o->shadow_rays, i, 1, NULL, &o->shadow_hits[i]);
}
ORC_EXPORT void orc_stage_accumulate(void* h)
{
    orc* o = (orc*)h;
    uint32_t n = o->width * o->height;
    for (uint32_t i = 0; i < n; ++i) AccumulateDirectSamples(o, i);
}
ORC_EXPORT void orc_stage_advance(void* h) { ((orc*)h)->sample_counter++; }
ORC_EXPORT void orc_stage_compute_aovs(void* h)
{
    orc* o = (orc*)h;
    uint32_t n = o->width * o->height;
    for (uint32_t i = 0; i < n; ++i) GenerateAOV(o, i);
}
ORC_EXPORT void orc_stage_denoise(void* h)
{
    orc* o = (orc*)h;
    uint32_t n = o->width * o->height;
    for (uint32_t i = 0; i < n; ++i) TemporalAccumulation(o, i);
}
ORC_EXPORT void orc_stage_copy_history(void* h)                         /* cl_pt_integrator.cpp:670-675 */
{
    orc* o = (orc*)h;
    size_t n = (size_t)o->width * o->height;
    memcpy(o->prev_radiance, o->radiance, n * sizeof(rt_float4));
    memcpy(o->prev_depth, o->depth, n * sizeof(float));
}

/* Integrator::Integrate(), integrator.cpp:27-59 */
ORC_EXPORT void orc_integrate(void* h)
{
    orc* o = (orc*)h;
    if (o->request_reset || o->denoiser) { orc_stage_reset(h); o->request_reset = 0; }
    orc_stage_generate_rays(h);
    for (uint32_t bounce = 0; bounce <= o->max_bounces; ++bounce)
    {
        orc_stage_intersect(h, bounce);
        if (bounce == 0) orc_stage_compute_aovs(h);
        orc_stage_shade_miss(h, bounce);
        orc_stage_clear_counters(h, bounce);
        orc_stage_shade_hits(h, bounce);
        orc_stage_intersect_shadow(h);
        orc_stage_accumulate(h);
        uint32_t active = o->ray_counter[bounce & 1];
        o->total_closest += active;
        o->total_shadow += o->shadow_ray_counter;
        if (bounce < 64) { o->last_active[bounce] = active; o->last_shadow[bounce] = o->shadow_ray_counter; }
    }
    orc_stage_advance(h);
    if (o->denoiser)
    {
        orc_stage_denoise(h);
        orc_stage_copy_history(h);
    }
}

/* ResolveRadiance, resolve_radiance.cl:31-86 */
ORC_EXPORT const float* orc_resolve(void* h)
{
    orc* o = (orc*)h;
    size_t n = (size_t)o->width * o->height;
    float spp = (float)o->sample_counter;
    for (size_t i = 0; i < n; ++i)
    {
        rt_float4* out = &o->resolved[i];
        out->w = 1.0f;
        if (o->aov == 1)        /* DIFFUSE_INDEX :52-56 */
        {
            out->x = o->diffuse_albedo[i].x; out->y = o->diffuse_albedo[i].y; out->z = o->diffuse_albedo[i].z;
        }
        else if (o->aov == 2)   /* DEPTH_INDEX :57-62 */
        {
            float d = o->depth[i] * 0.1f;
            out->x = d; out->y = d; out->z = d;
        }
        else if (o->aov == 3)   /* NORMAL_INDEX :63-68 */
        {
            out->x = o->normal[i].x * 0.5f + 0.5f; out->y = o->normal[i].y * 0.5f + 0.5f; out->z = o->normal[i].z * 0.5f + 0.5f;
        }
        else if (o->aov == 4)   /* MOTION_VECTORS_INDEX :69-73 */
        {
            out->x = o->velocity[i].x; out->y = o->velocity[i].y; out->z = 0.0f;
        }
        else                    /* shaded colour :76-85 */
        {
            float hx = o->radiance[i].x, hy = o->radiance[i].y, hz = o->radiance[i].z;
            if (!o->denoiser) { hx = hx / spp; hy = hy / spp; hz = hz / spp; }
            out->x = hx / (hx + 1.0f); out->y = hy / (hy + 1.0f); out->z = hz / (hz + 1.0f);
        }
    }
    return (const float*)o->resolved;
}

ORC_EXPORT const float* orc_radiance(void* h) { return (const float*)((orc*)h)->radiance; }
ORC_EXPORT uint32_t orc_sample_count(void* h) { return ((orc*)h)->sample_counter; }
ORC_EXPORT void orc_ray_totals(void* h, uint64_t* closest, uint64_t* shadow)
{
    *closest = ((orc*)h)->total_closest; *shadow = ((orc*)h)->total_shadow;
}
ORC_EXPORT void orc_last_counts(void* h, uint32_t* active, uint32_t* shadow, uint32_t n)
{
    orc* o = (orc*)h;
    for (uint32_t i = 0; i < n && i < 64; ++i) { active[i] = o->last_active[i]; shadow[i] = o->last_shadow[i]; }
}
/* stats[0..9] = closest_nodes, closest_tris, shadow_nodes, shadow_tris, escaped, hits, emissive, texels,
   unoccluded, outgoing (cumulative since creation) */
ORC_EXPORT void orc_stats(void* h, uint64_t* stats)
{
    orc* o = (orc*)h;
    stats[0] = o->closest_nodes; stats[1] = o->closest_tris; stats[2] = o->shadow_nodes; stats[3] = o->shadow_tris;
    stats[4] = o->n_escaped; stats[5] = o->n_hits; stats[6] = o->n_emissive; stats[7] = o->n_texels;
    stats[8] = o->n_unoccluded; stats[9] = o->n_outgoing;
}

ORC_EXPORT void* orc_buffer(void* h, const char* name)
{
    orc* o = (orc*)h;
    if (!strcmp(name, "rays0")) return o->rays[0];
    if (!strcmp(name, "rays1")) return o->rays[1];
    if (!strcmp(name, "pixel_indices0")) return o->pixel_indices[0];
    if (!strcmp(name, "pixel_indices1")) return o->pixel_indices[1];
    if (!strcmp(name, "ray_counter0")) return &o->ray_counter[0];
    if (!strcmp(name, "ray_counter1")) return &o->ray_counter[1];
    if (!strcmp(name, "shadow_rays")) return o->shadow_rays;
    if (!strcmp(name, "shadow_pixel_indices")) return o->shadow_pixel_indices;
    if (!strcmp(name, "shadow_ray_counter")) return &o->shadow_ray_counter;
    if (!strcmp(name, "hits")) return o->hits;
    if (!strcmp(name, "shadow_hits")) return o->shadow_hits;
    if (!strcmp(name, "throughputs")) return o->throughputs;
    if (!strcmp(name, "direct_light_samples")) return o->direct_light_samples;
    if (!strcmp(name, "radiance")) return o->radiance;
    if (!strcmp(name, "sample_counter")) return &o->sample_counter;
    return NULL;
}

/* ---- the 4-wide quantized traversal of the HIP path, restated on the CPU --------------------------------------
 * NOT a reference function: this is k_trace_w4 (raytracing_amd/csrc/trace_kernels.h) ray by ray, over the records
 * build_wide_bvh (rt_hip.hip) makes, with the kernel's own arithmetic -- exact dequantisation, ONE fma per plane with
 * the outward 2^-20 M margin, the per-octant order table, pops pre-culled by the stored entry distance, every leaf
 * re-tested with its exact bounds and the ray's current t_max -- so that tests/test_wide_traversal_oracle.py can show on
 * the CPU, against TraceOne above (trace_bvh.cl:99-211), that the wide walk reaches the reference's leaves in the
 * reference's order and returns the reference's hits bit for bit, and so that the walk's statistics (visits, pushes,
 * culled pops, stack depth) can be had without a GPU.  Rays the kernel hands to its BVH2 follow-up (non-finite or huge
 * 1/dir, far origins) take TraceOne here as well. */
typedef struct { float ox, oy, oz; uint32_t meta; uint32_t lo[3]; uint32_t hi[3]; uint32_t ref[4]; uint32_t order; uint32_t pad; } orc_wide_node;
#define ORC_LEAF_BIT 0x80000000u
#define ORC_EMPTY_REF 0xFFFFFFFFu

static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline float w_max3(float a, float b, float c) { return f_max(f_max(a, b), c); }
static inline float w_min3(float a, float b, float c) { return f_min(f_min(a, b), c); }

/* counters: 0 rays, 1 wide-node visits, 2 leaf arrivals, 3 leaf box tests failed, 4 triangle tests, 5 pushes,
 * 6 pops culled by their entry distance, 7 deepest stack, 8 rays left to the BVH2 walk, 9 slots that passed their box test */
/* events (optional): per ray up to `stride` bytes, one per step of the walk in order -- 'N' a wide-node visit, 'L' a leaf
 * arrival (exact box test + first triangle), 'T' a further triangle of that leaf -- and lengths[ray] = steps taken (may exceed
 * stride: the tail is not recorded); what tools/wave_schedule_model.py replays lane by lane */
static int wide_trace_impl(void* h, const void* wide_records, uint32_t n_wide, uint32_t entry_ref, const rt_ray* rays,
    uint32_t n_rays, int shadow, rt_hit* hits_out, uint32_t* shadow_out, uint64_t* counters, uint8_t* events, uint32_t stride,
    uint32_t* lengths)
{
    orc* o = (orc*)h;
    const int direct = (shadow & 2) != 0;                                /* bit 1 of `shadow`: the direct form of the walk */
    const int ordered_shadow = (shadow & 4) != 0;                        /* bit 2: shadow rays visit slots near-first too (analysis) */
    const int by_distance = (shadow & 8) != 0;                           /* bit 3: slots visited by entry distance instead of the record's
                                                                          * order table (analysis of the tolerance mode: tools/own_tree_study.py) */
    shadow &= 1;
    const orc_wide_node* wn = (const orc_wide_node*)wide_records;
    /* leaf ref (first triangle) -> the BVH2 leaf node that holds its exact bounds and primitive count */
    uint32_t* leaf_of = (uint32_t*)malloc(((size_t)o->n_triangles + 1) * sizeof(uint32_t));
    if (!leaf_of) return 1;
    memset(leaf_of, 0xFF, ((size_t)o->n_triangles + 1) * sizeof(uint32_t));
    for (uint32_t i = 0; i < o->n_nodes; ++i)
        if ((o->nodes[i].num_primitives_axis >> 16) != 0 && o->nodes[i].offset < o->n_triangles) leaf_of[o->nodes[i].offset] = i;
    const float INF = __builtin_inff();
    int rc = 0;
    for (uint32_t ri = 0; ri < n_rays && rc == 0; ++ri)
    {
        const rt_ray ray = rays[ri];
        const v3 org = V3(ray.origin.x, ray.origin.y, ray.origin.z), dir = V3(ray.direction.x, ray.direction.y, ray.direction.z);
        const float t_min = 0.0f;                                        /* the HIP queues carry no t_min: it is 0 */
        float t_max = ray.direction.w;
        const v3 inv = V3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
        const uint32_t sign_bits = (inv.x < 0.0f ? 1u : 0u) | (inv.y < 0.0f ? 2u : 0u) | (inv.z < 0.0f ? 4u : 0u);
        counters[0]++;
        uint32_t n_ev = 0;
#define ORC_EVENT(c) do { if (events && n_ev < stride) events[(size_t)ri * stride + n_ev] = (uint8_t)(c); ++n_ev; } while (0)
        if (lengths) lengths[ri] = 0;
        /* ray_inverse (kernels_common.h) + the origin test at the ray's start (k_trace_w4, phase A) */
        const float lim = 0x1p96f;
        const int slow = !(__builtin_fabsf(inv.x) < lim && __builtin_fabsf(inv.y) < lim && __builtin_fabsf(inv.z) < lim) ||
                         !(w_max3(__builtin_fabsf(org.x), __builtin_fabsf(org.y), __builtin_fabsf(org.z)) < 0x1p29f) || ray.origin.w != 0.0f;
        if (slow)
        {
            counters[8]++;
            rt_hit hh; uint32_t sh = RT_INVALID_ID;
            TraceOne(o, rays, ri, shadow, &hh, &sh);
            if (shadow) shadow_out[ri] = sh; else hits_out[ri] = hh;
            continue;
        }
        rt_hit hit; memset(&hit, 0, sizeof(hit)); hit.primitive_id = RT_INVALID_ID;
        uint32_t shadow_hit = RT_INVALID_ID;
        struct { uint32_t ref; float entry; } stack[128];
        int sp = 0;
        uint32_t ref = entry_ref;
        int have = 1;
        while (have)
        {
            if (ref & ORC_LEAF_BIT)
            {
                /* B: the leaf's exact box with the CURRENT t_max (the reference's RayBounds on the leaf node), then its
                 * triangles in array order */
                counters[2]++;
                ORC_EVENT('L');
                const uint32_t first = ref & ~ORC_LEAF_BIT;
                const uint32_t leaf = first <= o->n_triangles ? leaf_of[first] : RT_INVALID_ID;
                if (leaf == RT_INVALID_ID) { rc = 2; break; }
                int stop = 0;
                if (RayBounds(&o->nodes[leaf], org, inv, t_min, t_max))
                {
                    const uint32_t np = o->nodes[leaf].num_primitives_axis >> 16;
                    for (uint32_t i = 0; i < np && !stop; ++i)
                    {
                        counters[4]++;
                        if (i > 0) ORC_EVENT('T');
                        float u, v, t;
                        if (RayTriangle(org, dir, t_min, t_max, &o->triangles[first + i], &u, &v, &t))
                        {
                            hit.bc.x = u; hit.bc.y = v; hit.t = t; hit.primitive_id = first + i;
                            t_max = t;
                            if (shadow) { shadow_hit = 0; stop = 1; }
                        }
                    }
                }
                else counters[3]++;
                if (stop) break;
            }
            else
            {
                /* C: one wide node */
                if (ref >= n_wide) { rc = 3; break; }
                counters[1]++;
                ORC_EVENT('N');
                const orc_wide_node* n = &wn[ref];
                const float cx = u2f((n->meta & 0xFFu) << 23), cy = u2f(((n->meta >> 8) & 0xFFu) << 23), cz = u2f(((n->meta >> 16) & 0xFFu) << 23);
                const int nx = (sign_bits & 1u) != 0u, ny = (sign_bits & 2u) != 0u, nz = (sign_bits & 4u) != 0u;
                const uint32_t nwx = nx ? n->hi[0] : n->lo[0], fwx = nx ? n->lo[0] : n->hi[0];
                const uint32_t nwy = ny ? n->hi[1] : n->lo[1], fwy = ny ? n->lo[1] : n->hi[1];
                const uint32_t nwz = nz ? n->hi[2] : n->lo[2], fwz = nz ? n->lo[2] : n->hi[2];
                const float ax = cx * inv.x, ay = cy * inv.y, az = cz * inv.z;
                const float bx = (n->ox - org.x) * inv.x, by = (n->oy - org.y) * inv.y, bz = (n->oz - org.z) * inv.z;
                const float mx = __builtin_fmaf(255.0f, __builtin_fabsf(ax), __builtin_fabsf(bx)) + 0x1p-100f;
                const float my = __builtin_fmaf(255.0f, __builtin_fabsf(ay), __builtin_fabsf(by)) + 0x1p-100f;
                const float mz = __builtin_fmaf(255.0f, __builtin_fabsf(az), __builtin_fabsf(bz)) + 0x1p-100f;
                const float bnx = __builtin_fmaf(-0x1p-20f, mx, bx), bfx = __builtin_fmaf(0x1p-20f, mx, bx);
                const float bny = __builtin_fmaf(-0x1p-20f, my, by), bfy = __builtin_fmaf(0x1p-20f, my, by);
                const float bnz = __builtin_fmaf(-0x1p-20f, mz, bz), bfz = __builtin_fmaf(0x1p-20f, mz, bz);
                uint32_t r[4] = {n->ref[0], n->ref[1], n->ref[2], n->ref[3]};
                float e[4];
                for (int k = 0; k < 4; ++k)
                {
                    const float tnx = __builtin_fmaf((float)((nwx >> (8 * k)) & 0xFFu), ax, bnx);
                    const float tny = __builtin_fmaf((float)((nwy >> (8 * k)) & 0xFFu), ay, bny);
                    const float tnz = __builtin_fmaf((float)((nwz >> (8 * k)) & 0xFFu), az, bnz);
                    const float tfx = __builtin_fmaf((float)((fwx >> (8 * k)) & 0xFFu), ax, bfx);
                    const float tfy = __builtin_fmaf((float)((fwy >> (8 * k)) & 0xFFu), ay, bfy);
                    const float tfz = __builtin_fmaf((float)((fwz >> (8 * k)) & 0xFFu), az, bfz);
                    const float entry = f_max(w_max3(tnx, tny, tnz), t_min);
                    const float exit = f_min(w_min3(tfx, tfy, tfz), t_max);
                    e[k] = (exit >= entry && r[k] != ORC_EMPTY_REF) ? entry : INF;
                    if (e[k] < INF) counters[9]++;
                }
                if (by_distance)
                {
                    for (int i = 1; i < 4; ++i)
                        for (int j = i; j > 0 && e[j] < e[j - 1]; --j)
                        {
                            const uint32_t tr = r[j]; r[j] = r[j - 1]; r[j - 1] = tr;
                            const float te = e[j]; e[j] = e[j - 1]; e[j - 1] = te;
                        }
                }
                else if (!shadow || ordered_shadow)
                {
                    /* four conditional exchanges bring the occupied slots into the reference's visit order for this direction
                     * octant, whatever the shape of the BVH2 subtree the record folds (build_wide_bvh stores the slots where
                     * that is possible and tabulates the settings) */
                    const uint32_t sw = n->order >> (4u * (sign_bits & 7u));
                    uint32_t tr; float te;
#define ORC_SWAP(c, i, j) if (c) { tr = r[i]; r[i] = r[j]; r[j] = tr; te = e[i]; e[i] = e[j]; e[j] = te; }
                    ORC_SWAP(sw & 1u, 0, 1) ORC_SWAP(sw & 2u, 2, 3) ORC_SWAP(sw & 4u, 0, 2) ORC_SWAP(sw & 8u, 1, 3)
#undef ORC_SWAP
                }
                /* plain form: positions 3..1 wait on the stack, position 0 is visited next if it passed.  direct form
                 * (RT_OPT_TRACE_VARIANT 15): the FIRST passing position is visited next, only the later ones are pushed. */
                int first = 0;
                if (direct) { while (first < 4 && !(e[first] < INF)) ++first; }
                for (int k = 3; k >= first + 1; --k)
                    if (e[k] < INF)
                    {
                        if (sp >= 128) { rc = 4; break; }
                        stack[sp].ref = r[k]; stack[sp].entry = e[k]; ++sp; counters[5]++;
                        if ((uint64_t)sp > counters[7]) counters[7] = (uint64_t)sp;
                    }
                if (rc) break;
                if (first < 4 && e[first] < INF) { ref = r[first]; continue; }
            }
            /* pop: closest-hit rays skip entries whose (conservative) entry distance lies behind the current t_max */
            have = 0;
            while (sp > 0)
            {
                --sp;
                if (shadow || t_max >= stack[sp].entry) { ref = stack[sp].ref; have = 1; break; }
                counters[6]++;
            }
        }
        if (shadow) shadow_out[ri] = shadow_hit; else hits_out[ri] = hit;
        if (lengths) lengths[ri] = n_ev;
    }
#undef ORC_EVENT
    free(leaf_of);
    return rc;
}

ORC_EXPORT int orc_wide_trace(void* h, const void* wide_records, uint32_t n_wide, uint32_t entry_ref, const rt_ray* rays,
    uint32_t n_rays, int shadow, rt_hit* hits_out, uint32_t* shadow_out, uint64_t* counters)
{
    return wide_trace_impl(h, wide_records, n_wide, entry_ref, rays, n_rays, shadow, hits_out, shadow_out, counters, NULL, 0, NULL);
}

ORC_EXPORT int orc_wide_trace_events(void* h, const void* wide_records, uint32_t n_wide, uint32_t entry_ref, const rt_ray* rays,
    uint32_t n_rays, int shadow, rt_hit* hits_out, uint32_t* shadow_out, uint64_t* counters, uint8_t* events, uint32_t stride,
    uint32_t* lengths)
{
    return wide_trace_impl(h, wide_records, n_wide, entry_ref, rays, n_rays, shadow, hits_out, shadow_out, counters, events, stride, lengths);
}

/* ---- what would a W-wide tree cost?  (analysis only: tools/own_tree_study.py; nothing of the product is restated here) -------------
 * The SAH-optimal fold of a BVH2 -- the dynamic programme of build_wide_bvh (rt_hip.hip), for any width W -- and a walk of it with
 * EXACT boxes: a record visit tests its <= W slots, closest-hit rays take the passing ones nearest first, leaves cost one pass per
 * triangle.  counters: 0 rays, 1 record visits, 2 leaf arrivals, 3 triangle tests, 4 slots tested, 5 records in the tree.
 * `nodes` / `nn`: the BVH2 to fold (NULL: the oracle's own, i.e. the reference's); its leaves must be the oracle's leaves. */
typedef struct { const rt_bvh_node* nodes; uint32_t nn; uint32_t W; uint8_t* split; uint32_t* open; } nw_fold;
static int nw_leaf(const nw_fold* f, uint32_t i) { return (f->nodes[i].num_primitives_axis >> 16) != 0; }
static void nw_frontier(const nw_fold* f, uint32_t n, uint32_t k, uint32_t* out, uint32_t* count)
{
    const uint32_t c[2] = {n + 1, f->nodes[n].offset};
    const uint32_t give[2] = {f->split[(size_t)n * (f->W + 1) + k], k - f->split[(size_t)n * (f->W + 1) + k]};
    for (int i = 0; i < 2; ++i)
    {
        if (!nw_leaf(f, c[i]) && give[i] >= 2u && ((f->open[c[i]] >> give[i]) & 1u)) nw_frontier(f, c[i], give[i], out, count);
        else out[(*count)++] = c[i];
    }
}
/* counts[n] += 1 for every ray whose box test of BVH2 node n passes with the ray's INITIAL t_max: how often a record rooted at n
 * would be visited at most -- a measured visit probability to fold with instead of the surface area (analysis) */
ORC_EXPORT int orc_node_pass_counts(void* h, const rt_ray* rays, uint32_t n_rays, double* counts)
{
    orc* o = (orc*)h;
    const rt_bvh_node* nodes = o->nodes;
    for (uint32_t ri = 0; ri < n_rays; ++ri)
    {
        const rt_ray ray = rays[ri];
        const v3 org = V3(ray.origin.x, ray.origin.y, ray.origin.z), dir = V3(ray.direction.x, ray.direction.y, ray.direction.z);
        const v3 inv = V3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
        uint32_t stack[128]; int sp = 0;
        stack[sp++] = 0;
        while (sp > 0)
        {
            const uint32_t n = stack[--sp];
            if (!RayBounds(&nodes[n], org, inv, ray.origin.w, ray.direction.w)) continue;
            counts[n] += 1.0;
            if ((nodes[n].num_primitives_axis >> 16) != 0 || sp > 125) continue;
            stack[sp++] = nodes[n].offset;
            stack[sp++] = n + 1;
        }
    }
    return 0;
}

static int nwide_stats_impl(void* h, uint32_t W, const rt_bvh_node* nodes, uint32_t nn, const rt_ray* rays, uint32_t n_rays, int shadow, uint64_t* counters,
    const double* weights);
ORC_EXPORT int orc_nwide_stats(void* h, uint32_t W, const rt_bvh_node* nodes, uint32_t nn, const rt_ray* rays, uint32_t n_rays, int shadow, uint64_t* counters)
{
    return nwide_stats_impl(h, W, nodes, nn, rays, n_rays, shadow, counters, NULL);
}
/* the same with the fold's cost of a record = weights[root] instead of its surface area */
ORC_EXPORT int orc_nwide_stats_weighted(void* h, uint32_t W, const double* weights, const rt_ray* rays, uint32_t n_rays, int shadow, uint64_t* counters)
{
    return nwide_stats_impl(h, W, NULL, 0, rays, n_rays, shadow, counters, weights);
}
static int nwide_stats_impl(void* h, uint32_t W, const rt_bvh_node* nodes, uint32_t nn, const rt_ray* rays, uint32_t n_rays, int shadow, uint64_t* counters,
    const double* weights)
{
    orc* o = (orc*)h;
    if (W < 2 || W > 16) return 1;
    if (!nodes) { nodes = o->nodes; nn = o->n_nodes; }
    nw_fold f = {nodes, nn, W, NULL, NULL};
    f.split = (uint8_t*)calloc((size_t)nn * (W + 1), 1);
    f.open = (uint32_t*)calloc(nn, sizeof(uint32_t));
    double* T = (double*)calloc(nn, sizeof(double));
    double* F = (double*)calloc((size_t)nn * (W + 1), sizeof(double));
    if (!f.split || !f.open || !T || !F) return 2;
#define NW_G(c, i) (nw_leaf(&f, (c)) ? 0.0 : ((i) >= 2u ? (T[c] < F[(size_t)(c) * (W + 1) + (i)] ? T[c] : F[(size_t)(c) * (W + 1) + (i)]) : T[c]))
    for (uint32_t n = nn; n-- > 0;)
    {
        if (nw_leaf(&f, n)) continue;
        const uint32_t l = n + 1, r = nodes[n].offset;
        for (uint32_t k = 2; k <= W; ++k)
        {
            double best = 0.0; uint32_t at = 0;
            for (uint32_t i = 1; i < k; ++i)
            {
                const double c = NW_G(l, i) + NW_G(r, k - i);
                if (at == 0 || c < best) { best = c; at = i; }
            }
            F[(size_t)n * (W + 1) + k] = best;
            f.split[(size_t)n * (W + 1) + k] = (uint8_t)at;
        }
        const double dx = (double)nodes[n].bounds_max.x - nodes[n].bounds_min.x, dy = (double)nodes[n].bounds_max.y - nodes[n].bounds_min.y,
                     dz = (double)nodes[n].bounds_max.z - nodes[n].bounds_min.z;
        T[n] = (weights ? weights[n] : dx * dy + dy * dz + dz * dx) + F[(size_t)n * (W + 1) + W];
        for (uint32_t i = 2; i <= W; ++i)
            if (F[(size_t)n * (W + 1) + i] < T[n]) f.open[n] |= 1u << i;
    }
#undef NW_G
    /* records in the tree */
    {
        uint32_t* todo = (uint32_t*)malloc((size_t)nn * sizeof(uint32_t));
        uint32_t top = 0, recs = 0;
        if (!nw_leaf(&f, 0)) todo[top++] = 0;
        while (top)
        {
            const uint32_t n = todo[--top];
            ++recs;
            uint32_t slots[16], cnt = 0;
            nw_frontier(&f, n, W, slots, &cnt);
            for (uint32_t k = 0; k < cnt; ++k) if (!nw_leaf(&f, slots[k])) todo[top++] = slots[k];
        }
        counters[5] = recs;
        free(todo);
    }
    for (uint32_t ri = 0; ri < n_rays; ++ri)
    {
        const rt_ray ray = rays[ri];
        const v3 org = V3(ray.origin.x, ray.origin.y, ray.origin.z), dir = V3(ray.direction.x, ray.direction.y, ray.direction.z);
        const v3 inv = V3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
        const float t_min = ray.origin.w;
        float t_max = ray.direction.w;
        counters[0]++;
        struct { uint32_t node; float entry; } stack[256];
        int sp = 0, done = 0;
        if (RayBounds(&nodes[0], org, inv, t_min, t_max)) { stack[sp].node = 0; stack[sp].entry = t_min; ++sp; }
        while (sp > 0 && !done)
        {
            --sp;
            const uint32_t n = stack[sp].node;
            if (!shadow && stack[sp].entry > t_max) continue;
            if (nw_leaf(&f, n))
            {
                counters[2]++;
                const uint32_t first = nodes[n].offset, np = nodes[n].num_primitives_axis >> 16;
                for (uint32_t i = 0; i < np && !done; ++i)
                {
                    counters[3]++;
                    float u, v, t;
                    if (RayTriangle(org, dir, t_min, t_max, &o->triangles[first + i], &u, &v, &t)) { t_max = t; if (shadow) done = 1; }
                }
                continue;
            }
            counters[1]++;
            uint32_t slots[16], cnt = 0;
            nw_frontier(&f, n, W, slots, &cnt);
            uint32_t pr[16]; float pe[16]; int np = 0;
            for (uint32_t k = 0; k < cnt; ++k)
            {
                counters[4]++;
                const rt_bvh_node* b = &nodes[slots[k]];
                /* the slab test of RayBounds, keeping the entry distance */
                const float t0x = (b->bounds_min.x - org.x) * inv.x, t1x = (b->bounds_max.x - org.x) * inv.x;
                const float t0y = (b->bounds_min.y - org.y) * inv.y, t1y = (b->bounds_max.y - org.y) * inv.y;
                const float t0z = (b->bounds_min.z - org.z) * inv.z, t1z = (b->bounds_max.z - org.z) * inv.z;
                const float en = f_max(f_max(f_max(f_min(t0x, t1x), f_min(t0y, t1y)), f_min(t0z, t1z)), t_min);
                const float ex = f_min(f_min(f_min(f_max(t0x, t1x), f_max(t0y, t1y)), f_max(t0z, t1z)), t_max);
                if (ex >= en) { pr[np] = slots[k]; pe[np] = en; ++np; }
            }
            if (!shadow)
                for (int i = 1; i < np; ++i)
                    for (int j = i; j > 0 && pe[j] < pe[j - 1]; --j)
                    {
                        const uint32_t tr = pr[j]; pr[j] = pr[j - 1]; pr[j - 1] = tr;
                        const float te = pe[j]; pe[j] = pe[j - 1]; pe[j - 1] = te;
                    }
            for (int i = np - 1; i >= 0 && sp < 256; --i) { stack[sp].node = pr[i]; stack[sp].entry = pe[i]; ++sp; }
        }
    }
    free(f.split); free(f.open); free(T); free(F);
    return 0;
}

/* known-answer access to the leaf functions (tests/test_oracle_kat.py) */
ORC_EXPORT uint32_t orc_wang_hash(uint32_t x) { return WangHash(x); }
ORC_EXPORT float orc_sample_random(uint32_t x, uint32_t y, uint32_t s, uint32_t b, uint32_t t)
{
    return SampleRandom(x, y, s, b, t);
}
ORC_EXPORT float orc_tanf(float x) { return rt_tanf(x); }
ORC_EXPORT float orc_sinf(float x) { return rt_sinf(x); }
ORC_EXPORT float orc_cosf(float x) { return rt_cosf(x); }
ORC_EXPORT float orc_powf(float x, float y) { return rt_powf(x, y); }
ORC_EXPORT float orc_atan2f(float y, float x) { return rt_atan2f(y, x); }
ORC_EXPORT float orc_acosf(float x) { return rt_acosf(x); }
