// aov_kernels.h -- AOV generation, temporal denoiser, resolve (aov.cl, denoiser.cl,
// resolve_radiance.cl).
#pragma once
#include "kernels_common.h"

// ---------------------------------------------------------------------------
// AOVs, temporal denoiser, resolve (aov.cl, denoiser.cl, resolve_radiance.cl)
// Interactive per-frame features: one sample in flight, whole image on one GPU.
// ---------------------------------------------------------------------------
struct DAov
{
    float4* diffuse_albedo;   // float3 in the reference (16 B)
    float* depth;
    float4* normal;
    float2* velocity;
};

// the AOV resets of RayGeneration (raygeneration.cl:129-132)
__global__ __launch_bounds__(256) void k_aov_clear(DAov aov, uint32_t n)
{
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    aov.diffuse_albedo[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    aov.depth[i] = RT_MAX_RENDER_DIST;
    aov.normal[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    aov.velocity[i] = make_float2(0.0f, 0.0f);
}

RT_DEV f2 ProjectScreen(f3 position, const rt_camera& cam, float tan_half_fov)      // aov.cl:30-42
{
    f3 cpos = F3(cam.position.x, cam.position.y, cam.position.z);
    f3 front = F3(cam.front.x, cam.front.y, cam.front.z), up = F3(cam.up.x, cam.up.y, cam.up.z);
    f3 d = normalize3(position - cpos);
    f3 ipd = d / dot3(front, d);
    float angle = tan_half_fov;
    f3 right = cross3(front, up);
    float u = dot3(right, ipd) / (angle * cam.aspect_ratio);
    float v = dot3(up, ipd) / (angle);
    f2 r;
    r.x = u * 0.5f + 0.5f;
    r.y = v * 0.5f + 0.5f;
    return r;
}

// GenerateAOV, aov.cl:44-110 (first-hit albedo / depth / normal / screen-space velocity)
__global__ __launch_bounds__(256) void k_aov(DScene sc, const float4* __restrict__ o4, const float4* __restrict__ d4,
    const float4* __restrict__ hits, const uint32_t* __restrict__ count_ptr, rt_camera cam, rt_camera prev_cam,
    float tan_cam, float tan_prev, DAov aov)
{
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= *count_ptr) return;
    float4 hit = hits[i];
    uint32_t prim = __float_as_uint(hit.z);
    if (prim == RT_INVALID_ID) return;
    float4 ro = o4[i];
    uint32_t pix = __float_as_uint(d4[i].w);                               // one sample in flight: id == pixel
    const float4* tp = sc.tris_sh + (size_t)prim * 8;
    float4 q0 = tp[0], q1 = tp[1], q2 = tp[2], q3 = tp[3], q4 = tp[4], q5 = tp[5], q6 = tp[6];
    f3 p1 = xyz(q0), p2 = xyz(q1), p3 = xyz(q2);
    f3 n1 = xyz(q3), n2 = xyz(q4), n3 = xyz(q5);
    float bu = hit.x, bv = hit.y;
    float w0 = 1.0f - bu - bv;
    f3 position = p1 * w0 + p2 * bu + p3 * bv;
    f2 texcoord;
    texcoord.x = q0.w * w0 + q2.w * bu + q4.w * bv;
    texcoord.y = q1.w * w0 + q3.w * bu + q5.w * bv;
    f3 normal = normalize3(n1 * w0 + n2 * bu + n3 * bv);
    Material material;
    ApplyTextures(sc, __float_as_uint(q6.x), material, texcoord);
    aov.diffuse_albedo[pix] = make_float4(material.diffuse_albedo.x, material.diffuse_albedo.y, material.diffuse_albedo.z, 0.0f);
    aov.depth[pix] = length3(F3(ro.x, ro.y, ro.z) - position);
    aov.normal[pix] = make_float4(normal.x, normal.y, normal.z, 0.0f);
    f2 a = ProjectScreen(position, cam, tan_cam), b = ProjectScreen(position, prev_cam, tan_prev);
    aov.velocity[pix] = make_float2(a.x - b.x, a.y - b.y);
}

// TemporalAccumulation, denoiser.cl:27-79: reproject, depth test, mix(cur, prev, 0.9)
__global__ __launch_bounds__(256) void k_denoise(uint32_t width, uint32_t height, float4* __restrict__ radiance,
    const float4* __restrict__ prev_radiance, const float* __restrict__ depth, const float* __restrict__ prev_depth,
    const float2* __restrict__ velocity)
{
    uint32_t pixel_idx = blockIdx.x * 256u + threadIdx.x;
    int x = (int)(pixel_idx % width);
    int y = (int)(pixel_idx / width);
    if ((uint32_t)x >= width || (uint32_t)y >= height) return;
    float depth_value = depth[pixel_idx];
    if (depth_value == RT_MAX_RENDER_DIST) return;                         // background
    float2 motion = velocity[pixel_idx];
    float prev_u = ((float)x + 0.5f) / (float)width - motion.x;
    float prev_v = ((float)y + 0.5f) / (float)height - motion.y;
    int prev_x = (int)(prev_u * (float)width);
    int prev_y = (int)(prev_v * (float)height);
    if (prev_x < 0 || (uint32_t)prev_x >= width || prev_y < 0 || (uint32_t)prev_y >= height) return;
    int prev_idx = prev_y * (int)width + prev_x;
    float prev_depth_value = prev_depth[prev_idx];
    if (__builtin_fabsf(depth_value - prev_depth_value) / depth_value > 0.1f) return;   // depth similarity
    float4 cur = radiance[pixel_idx];
    float4 prev = prev_radiance[prev_idx];
    f3 m = mix3(F3(cur.x, cur.y, cur.z), F3(prev.x, prev.y, prev.z), 0.9f);
    radiance[pixel_idx] = make_float4(m.x, m.y, m.z, cur.w);
}

// ResolveRadiance, resolve_radiance.cl:31-86: AOV switch, else average + Reinhard
__global__ __launch_bounds__(256) void k_resolve(const float4* __restrict__ radiance, DAov aov, float4* __restrict__ out,
    uint32_t n, uint32_t sample_count, uint32_t aov_index, uint32_t denoiser)
{
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    if (aov_index == 1)
    {
        float4 v = aov.diffuse_albedo[i];
        out[i] = make_float4(v.x, v.y, v.z, 1.0f);
    }
    else if (aov_index == 2)
    {
        float d = aov.depth[i] * 0.1f;
        out[i] = make_float4(d, d, d, 1.0f);
    }
    else if (aov_index == 3)
    {
        float4 v = aov.normal[i];
        out[i] = make_float4(v.x * 0.5f + 0.5f, v.y * 0.5f + 0.5f, v.z * 0.5f + 0.5f, 1.0f);
    }
    else if (aov_index == 4)
    {
        float2 v = aov.velocity[i];
        out[i] = make_float4(v.x, v.y, 0.0f, 1.0f);
    }
    else
    {
        float4 r = radiance[i];
        float hx = r.x, hy = r.y, hz = r.z;
        if (!denoiser)                                                     // -D ENABLE_DENOISER: no division
        {
            float spp = (float)sample_count;
            hx = hx / spp; hy = hy / spp; hz = hz / spp;
        }
        out[i] = make_float4(hx / (hx + 1.0f), hy / (hy + 1.0f), hz / (hz + 1.0f), 1.0f);
    }
}
