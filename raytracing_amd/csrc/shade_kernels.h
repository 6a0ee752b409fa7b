// shade_kernels.h -- miss + surface shading + queue compaction (miss.cl, hit_surface.cl,
// material.h, bxdf.h, light.h, sampling.h) and the replay of the radiance log (k_flush).
#pragma once
#include "kernels_common.h"

// ---------------------------------------------------------------------------
// shading (miss.cl + hit_surface.cl + material.h + bxdf.h + light.h)
// ---------------------------------------------------------------------------
struct Material
{
    f3 diffuse_albedo; float roughness;
    f3 specular_albedo; float metalness;
    f3 emission; float ior; float transparency;
};

// material.h:251-264 + utils.h:123-131
RT_DEV f3 SampleTexture(const DScene& sc, uint32_t tex_idx, f2 uv)
{
    rt_texture tex = sc.textures[tex_idx];
    uv.x -= __builtin_floorf(uv.x);
    uv.y -= __builtin_floorf(uv.y);
    uv.y = 1.f - uv.y;
    int texel_x = cl_clampi((int)(uv.x * (float)tex.width), 0, tex.width - 1);
    int texel_y = cl_clampi((int)(uv.y * (float)tex.height), 0, tex.height - 1);
    int texel_addr = tex.data_start + texel_y * tex.width + texel_x;
    uint32_t data = sc.texture_data[texel_addr];
    float r = (float)(data & 0xFF) / 255.0f;
    float g = (float)((data >> 8) & 0xFF) / 255.0f;
    float b = (float)((data >> 16) & 0xFF) / 255.0f;
    return F3(cl_min(cl_max(r, 0.0f), 1.0f), cl_min(cl_max(g, 0.0f), 1.0f), cl_min(cl_max(b, 0.0f), 1.0f));
}

// pow(SampleTexture(...), 2.2f) (material.h:327,336,361).  A texel channel is one of 256 values,
// so the three rt_powf evaluations (~200 fp64 operations each, paid by the whole wave as soon as
// one lane has a textured material) are a table of the very same function, filled on the device
// by the very same code (k_fill_gamma_lut) -- identical bits by construction.
RT_DEV f3 SampleTextureGamma(const DScene& sc, uint32_t tex_idx, f2 uv)
{
    rt_texture tex = sc.textures[tex_idx];
    uv.x -= __builtin_floorf(uv.x);
    uv.y -= __builtin_floorf(uv.y);
    uv.y = 1.f - uv.y;
    int texel_x = cl_clampi((int)(uv.x * (float)tex.width), 0, tex.width - 1);
    int texel_y = cl_clampi((int)(uv.y * (float)tex.height), 0, tex.height - 1);
    uint32_t data = sc.texture_data[tex.data_start + texel_y * tex.width + texel_x];
    return F3(sc.gamma_lut[data & 0xFF], sc.gamma_lut[(data >> 8) & 0xFF], sc.gamma_lut[(data >> 16) & 0xFF]);
}

__global__ void k_fill_gamma_lut(float* __restrict__ lut)
{
    float v = (float)threadIdx.x / 255.0f;
    lut[threadIdx.x] = rt_powf(cl_min(cl_max(v, 0.0f), 1.0f), 2.2f);   // SampleTexture's clamp, then pow
}

RT_DEV f3 UnpackRGBTex(uint32_t data, uint32_t& idx)                    // utils.h:133-147
{
    float r = (float)(data & 0xFF), g = (float)((data >> 8) & 0xFF), b = (float)((data >> 16) & 0xFF);
    idx = (data >> 24) & 0xFF;
    return F3(r / 255.0f, g / 255.0f, b / 255.0f);
}

// mtl: the material's index (for the wide texture indices of rt_scene_desc::material_texture_indices, when given)
RT_DEV void ApplyTextures(const DScene& sc, uint32_t mtl, Material& out, f2 uv)   // material.h:319-369
{
    const rt_packed_material in = sc.materials[mtl];
    // texture index of slot k: the packed 8-bit field (0xFF = none), or the 16-bit side table (0xFFFF = none)
    const uint16_t* wide = sc.mat_tex16 ? sc.mat_tex16 + (size_t)mtl * 6u : nullptr;
    const uint32_t none = wide ? 0xFFFFu : RT_INVALID_TEXTURE_IDX;
    uint32_t idx;
    out.diffuse_albedo = UnpackRGBTex(in.diffuse_albedo, idx);
    if (wide) idx = wide[0];
    if (idx != none) out.diffuse_albedo = SampleTextureGamma(sc, idx, uv);
    out.specular_albedo = UnpackRGBTex(in.specular_albedo, idx);
    if (wide) idx = wide[1];
    if (idx != none) out.specular_albedo = SampleTextureGamma(sc, idx, uv);
    {
        uint32_t rgbe = in.emission;                                     // utils.h:149-158
        int r = (int)(rgbe & 0xFF), g = (int)((rgbe >> 8) & 0xFF), b = (int)((rgbe >> 16) & 0xFF);
        int e = (int)(rgbe >> 24);
        float f = rt_ldexpf(1.0f, e - (128 + 8));
        out.emission = F3((float)r * f, (float)g * f, (float)b * f);
    }
    uint32_t d = in.roughness_metalness;                                 // utils.h:160-174
    out.roughness = (float)(d & 0xFF) / 255.0f;
    uint32_t roughness_idx = wide ? wide[2] : (d >> 8) & 0xFF;
    out.metalness = (float)((d >> 16) & 0xFF) / 255.0f;
    uint32_t metalness_idx = wide ? wide[3] : (d >> 24) & 0xFF;
    if (roughness_idx != none) out.roughness = SampleTexture(sc, roughness_idx, uv).x;
    if (metalness_idx != none) out.metalness = SampleTexture(sc, metalness_idx, uv).x;
    d = in.ior_emission_idx_transparency;                                // utils.h:176-190
    out.ior = (float)(d & 0xFF) / 25.5f;
    uint32_t emission_idx = wide ? wide[4] : (d >> 8) & 0xFF;
    out.transparency = (float)((d >> 16) & 0xFF) / 255.0f;
    uint32_t transparency_idx = wide ? wide[5] : (d >> 24) & 0xFF;
    if (emission_idx != none)
        out.emission = out.emission * SampleTextureGamma(sc, emission_idx, uv);
    if (transparency_idx != none)
        out.transparency *= SampleTexture(sc, transparency_idx, uv).x;
}

RT_DEV float IorToF0(float ior_incident, float ior_transmitted)          // bxdf.h:57-61
{
    float result = (ior_transmitted - ior_incident) / (ior_transmitted + ior_incident);
    return result * result;
}

RT_DEV f3 FresnelSchlick(f3 f0, float h_dot_o)                           // bxdf.h:71-74
{
    float p = rt_powf(1.0f - h_dot_o, 5.0f);
    return F3(f0.x + (1.0f - f0.x) * p, f0.y + (1.0f - f0.y) * p, f0.z + (1.0f - f0.z) * p);
}

RT_DEV float GGX_D(float alpha, float n_dot_h)                           // bxdf.h:90-95
{
    float alpha2 = alpha * alpha;
    float denom = n_dot_h * n_dot_h * (alpha2 - 1.0f) + 1.0f;
    return alpha2 * RT_INV_PI / (denom * denom);
}

RT_DEV float V_SmithGGXCorrelated(float n_dot_i, float n_dot_o, float alphaG)   // bxdf.h:104-119
{
    float alphaG2 = alphaG * alphaG;
    float Lambda_GGXV = n_dot_o * __builtin_sqrtf((-n_dot_i * alphaG2 + n_dot_i) * n_dot_i + alphaG2);
    float Lambda_GGXL = n_dot_i * __builtin_sqrtf((-n_dot_o * alphaG2 + n_dot_o) * n_dot_o + alphaG2);
    return 0.5f / (Lambda_GGXV + Lambda_GGXL);
}

RT_DEV float Luma(f3 rgb) { return rgb.x * 0.299f + rgb.y * 0.587f + rgb.z * 0.114f; }   // utils.h:108-111

RT_DEV void tangent_frame(f3 n, f3& t, f3& b)                            // utils.h:101-103, bxdf.h:163-165
{
    f3 axis = __builtin_fabsf(n.x) > 0.001f ? F3(0.0f, 1.0f, 0.0f) : F3(1.0f, 0.0f, 0.0f);
    t = normalize3(cross3(axis, n));
    b = cross3(n, t);
}

RT_DEV f3 reflect3(f3 v, f3 n) { return v - n * (2.0f * dot3(v, n)); }   // utils.h:83-86

RT_DEV f3 EvaluateMaterial(const Material& m, f3 normal, f3 incoming, f3 outgoing)   // material.h:132-169
{
    if ((double)m.transparency < 0.5) return F3s(0.0f);
    f3 half_vec = normalize3(incoming + outgoing);
    float n_dot_i = cl_max(dot3(normal, incoming), RT_EPS);
    float n_dot_o = cl_max(dot3(normal, outgoing), RT_EPS);
    float n_dot_h = cl_max(dot3(normal, half_vec), RT_EPS);
    float h_dot_o = cl_max(dot3(half_vec, outgoing), RT_EPS);
    float alpha = m.roughness * m.roughness;
    float f0_dielectric = IorToF0(1.0f, m.ior);
    f3 f0 = mix3(F3s(f0_dielectric), m.specular_albedo, m.metalness);
    f3 diffuse_color = m.diffuse_albedo * (1.0f - m.metalness);
    f3 fresnel = FresnelSchlick(f0, h_dot_o);
    float specular = GGX_D(alpha, n_dot_h) * V_SmithGGXCorrelated(n_dot_i, n_dot_o, alpha);
    f3 diffuse = diffuse_color * RT_INV_PI;
    return F3(fresnel.x * specular + (1.0f - fresnel.x) * diffuse.x,
              fresnel.y * specular + (1.0f - fresnel.y) * diffuse.y,
              fresnel.z * specular + (1.0f - fresnel.z) * diffuse.z);
}

// material.h:171-241 with SampleSpecular :66-103, SampleDiffuse :51-64, SampleTransparency :105-117
template <bool FURNACE>
RT_DEV f3 SampleBxdf(float s1, f2 s, Material material, f3 normal, f3 incoming, f3& outgoing, float& pdf,
    float& offset, bool& delta /* the event chosen has a delta distribution (RT_SCENE_EMISSIVE_NEE) */)
{
    delta = false;
    if (FURNACE)
    {
        material.diffuse_albedo = F3s(1.0f);
        material.specular_albedo = F3s(1.0f);
    }
    float alpha = material.roughness * material.roughness;
    float f0_dielectric = IorToF0(1.0f, material.ior);
    f3 f0 = mix3(F3s(f0_dielectric), material.specular_albedo, material.metalness);
    f3 diffuse_albedo = material.diffuse_albedo * (1.0f - material.metalness);
    f3 specular_albedo = mix3(material.specular_albedo, F3s(1.0f), material.metalness);
    f3 fresnel = FresnelSchlick(f0, dot3(normal, incoming)) * specular_albedo;
    float specular_weight = Luma(specular_albedo * fresnel);
    float diffuse_weight = Luma(diffuse_albedo * F3(1.0f - fresnel.x, 1.0f - fresnel.y, 1.0f - fresnel.z));
    float weight_sum = diffuse_weight + specular_weight;
    float specular_sampling_pdf = specular_weight / weight_sum;
    float diffuse_sampling_pdf = diffuse_weight / weight_sum;

    offset = 1.0f;
    if ((double)material.transparency < 0.5)
    {
        pdf = 1.0f;
        outgoing = -incoming;
        offset = -1.0f;
        delta = true;
        return F3s(1.0f);
    }

    // both lobes turn the same angle phi = 2 pi s.x (bxdf.h:36,159): one sincos for the wave instead of one per branch
    double sd, cd;
    rtd_sincos((double)(RT_TWO_PI * s.x), &sd, &cd);
    f3 bxdf;
    if (s1 <= specular_sampling_pdf)
    {
        float spec;
        if (alpha <= 1e-4f)
        {
            outgoing = reflect3(-incoming, normal);
            pdf = 1.0f;
            float n_dot_o = dot3(outgoing, normal);
            spec = 1.0f / n_dot_o;
            delta = true;
        }
        else
        {
            // GGX_Sample bxdf.h:157-168 (fp64 literals in the reference -> fp64 divide + sqrt)
            float cos_theta = (float)(1.0 / __builtin_sqrt(1.0 + (double)(alpha * alpha * s.y) / (1.0 - (double)s.y)));
            float sin_theta = __builtin_sqrtf(cl_max(0.0f, 1.0f - cos_theta * cos_theta));
            f3 t, b;
            tangent_frame(normal, t, b);
            float cp = (float)cd, sn = (float)sd;
            f3 wh = normalize3(b * cp * sin_theta + t * sn * sin_theta + normal * cos_theta);
            outgoing = reflect3(-incoming, wh);
            float n_dot_o = dot3(normal, outgoing);
            float n_dot_h = dot3(normal, wh);
            float n_dot_i = dot3(normal, incoming);
            float D = GGX_D(alpha, n_dot_h);
            float G = V_SmithGGXCorrelated(n_dot_i, n_dot_o, alpha);
            pdf = D * n_dot_h / (4.0f * dot3(wh, outgoing));
            spec = D * G;
        }
        float m = cl_max(dot3(outgoing, normal), 0.0f);
        bxdf = F3(fresnel.x * spec * m, fresnel.y * spec * m, fresnel.z * spec * m);
        pdf *= specular_sampling_pdf;
    }
    else
    {
        // SampleHemisphereCosine bxdf.h:33-54 + TangentToWorld utils.h:99-106
        float sin_theta = __builtin_sqrtf(s.y);
        float cos_theta = __builtin_sqrtf(1.0f - s.y);
        pdf = cos_theta * RT_INV_PI;
        f3 tbn = F3((float)cd * sin_theta, (float)sd * sin_theta, cos_theta);
        f3 t, b;
        tangent_frame(normal, t, b);
        outgoing = normalize3(b * tbn.x + t * tbn.y + normal * tbn.z);
        f3 d = diffuse_albedo * RT_INV_PI;
        float m = cl_max(dot3(outgoing, normal), 0.0f);
        bxdf = F3((1.0f - fresnel.x) * d.x * m, (1.0f - fresnel.y) * d.y * m, (1.0f - fresnel.z) * d.z * m);
        pdf *= diffuse_sampling_pdf;
    }
    return bxdf;
}

// miss.cl:28-39 with the OpenCL 1.2 (8.2) linear / repeat / normalized sampler
RT_DEV f3 SampleSky(const DScene& sc, f3 dir)
{
    float cx = rt_atan2f(dir.x, dir.y) + RT_PI;
    float cy = rt_acosf(dir.z);
    cx = cx < 0.0f ? cx + RT_TWO_PI : cx;
    cx *= RT_INV_TWO_PI;
    cy *= RT_INV_PI;
    int w = sc.env_w, h = sc.env_h;
    float u = (cx - __builtin_floorf(cx)) * (float)w;
    float v = (cy - __builtin_floorf(cy)) * (float)h;
    float fu = __builtin_floorf(u - 0.5f);
    float fv = __builtin_floorf(v - 0.5f);
    // NaN coordinates (acos of a component one ulp above 1: OpenCL leaves the read undefined): texel (0, 0), NaN weights, a NaN
    // sample -- stated here instead of left to what v_cvt_i32_f32 does with a NaN; the shim under the reference's kernels and
    // oracle.c define the same (x86's conversion gave INT_MIN there, an address outside the image)
    int i0 = fu != fu ? 0 : (int)fu, j0 = fv != fv ? 0 : (int)fv;
    int i1 = i0 + 1, j1 = j0 + 1;
    if (i0 < 0) i0 = w + i0;
    if (i1 > w - 1) i1 = i1 - w;
    if (j0 < 0) j0 = h + j0;
    if (j1 > h - 1) j1 = j1 - h;
    float a = (u - 0.5f) - fu;
    float b = (v - 0.5f) - fv;
    float wa0 = 1.0f - a, wb0 = 1.0f - b;
    float4 t00 = sc.env[(size_t)j0 * w + i0];
    float4 t10 = sc.env[(size_t)j0 * w + i1];
    float4 t01 = sc.env[(size_t)j1 * w + i0];
    float4 t11 = sc.env[(size_t)j1 * w + i1];
    float w00 = wa0 * wb0, w10 = a * wb0, w01 = wa0 * b, w11 = a * b;
    return F3(w00 * t00.x + w10 * t10.x + w01 * t01.x + w11 * t11.x,
              w00 * t00.y + w10 * t10.y + w01 * t01.y + w11 * t11.y,
              w00 * t00.z + w10 * t10.z + w01 * t01.z + w11 * t11.z);
}

// Stream compaction for the two output queues: wave64 ballot + prefix inside a
// wave, LDS prefix across the waves of a block, ONE global atomic per block per
// queue -- instead of the reference's one same-address atomic per ray
// (hit_surface.cl:138,173).  Same-address L2 atomics retire at ~10 ns each on
// MI355X, so at ~20 M rays per launch even one atomic per wave (600 k of them) was
// the shade kernel's bottleneck; per 512-thread block it is 8x fewer.
#ifndef RT_SHADE_WAVES            // waves per SIMD k_shade is compiled for (78 VGPRs at 6; tools/build_variants.py: 7 / 8 tried in round 5)
#define RT_SHADE_WAVES 6
#endif
#ifndef RT_SHADE_BLOCK            // 256 measured no different (profiles/r02_variants_ab_call41.log)
#define RT_SHADE_BLOCK 512
#endif
RT_DEV void block_append2(bool want_a, bool want_b, uint32_t* counter_a, uint32_t* counter_b, uint32_t& idx_a,
    uint32_t& idx_b)
{
    __shared__ uint32_t s_cnt[2][RT_SHADE_BLOCK / 64];
    __shared__ uint32_t s_base[2];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    unsigned long long ma = __ballot(want_a), mb = __ballot(want_b);
    if (lane == 0)
    {
        s_cnt[0][wave] = (uint32_t)__popcll(ma);
        s_cnt[1][wave] = (uint32_t)__popcll(mb);
    }
    __syncthreads();
    if (threadIdx.x < 2)
    {
        uint32_t total = 0;
        for (uint32_t w = 0; w < RT_SHADE_BLOCK / 64; ++w) total += s_cnt[threadIdx.x][w];
        s_base[threadIdx.x] = total ? atomicAdd(threadIdx.x == 0 ? counter_a : counter_b, total) : 0u;
    }
    __syncthreads();
    uint32_t pa = s_base[0], pb = s_base[1];
    for (uint32_t w = 0; w < wave; ++w) { pa += s_cnt[0][w]; pb += s_cnt[1][w]; }
    idx_a = pa + (uint32_t)__popcll(ma & lt);
    idx_b = pb + (uint32_t)__popcll(mb & lt);
}

// The same, with each block's entries grouped by a 3-bit key (the direction octant of the ray): a wave of the next
// trace launch then holds rays that start near each other AND point into the same octant.  Queue order is free: results
// are replayed per path (k_flush), counters count.
template <uint32_t KEYS>
RT_DEV void block_append2_keyed(bool want_a, uint32_t key_a, bool want_b, uint32_t key_b, uint32_t* counter_a, uint32_t* counter_b,
    uint32_t& idx_a, uint32_t& idx_b)
{
    constexpr uint32_t W = RT_SHADE_BLOCK / 64, N = KEYS * W;                  // (key, wave) cells per queue, key-major
    static_assert(2u * N <= RT_SHADE_BLOCK, "one thread per cell");
    __shared__ uint32_t s_cnt[2][KEYS][W];                                    // counts, then their exclusive prefix
    __shared__ uint32_t s_base[2];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t rank_a = 0, rank_b = 0;
#pragma unroll
    for (uint32_t k = 0; k < KEYS; ++k)
    {
        const unsigned long long ma = __ballot(want_a && key_a == k), mb = __ballot(want_b && key_b == k);
        if (lane == 0) { s_cnt[0][k][wave] = (uint32_t)__popcll(ma); s_cnt[1][k][wave] = (uint32_t)__popcll(mb); }
        if (key_a == k) rank_a = (uint32_t)__popcll(ma & lt);
        if (key_b == k) rank_b = (uint32_t)__popcll(mb & lt);
    }
    __syncthreads();
    uint32_t mine = 0, before = 0;
    const uint32_t q = threadIdx.x / N, cell = threadIdx.x % N;
    if (threadIdx.x < 2u * N)
    {
        const uint32_t* c = &s_cnt[q][0][0];
        mine = c[cell];
        for (uint32_t i = 0; i < cell; ++i) before += c[i];
    }
    __syncthreads();
    if (threadIdx.x < 2u * N)
    {
        (&s_cnt[q][0][0])[cell] = before;
        if (cell == N - 1u)
        {
            const uint32_t total = before + mine;
            s_base[q] = total ? atomicAdd(q == 0 ? counter_a : counter_b, total) : 0u;
        }
    }
    __syncthreads();
    idx_a = s_base[0] + s_cnt[0][key_a % KEYS][wave] + rank_a;
    idx_b = s_base[1] + s_cnt[1][key_b % KEYS][wave] + rank_b;
}

struct ShadeArgs
{
    const float4* in_o4; const float4* in_d4; const float4* in_thr; const float4* hits;
    float4* out_o4; float4* out_d4; float4* out_thr;
    float4* sh_o4; float4* sh_d4; uint32_t* sh_aux;   // sh_aux[i]: radiance-log entry of shadow ray i's deferred direct sample
    DLog log;                         // radiance log (kernels_common.h): entries, counts, overflow blocks
    const uint8_t* bn_sobol; const uint8_t* bn_scramble; const uint8_t* bn_rank;   // SamplerType::kBlueNoise tables
    DCounters* counters;
    uint32_t bounce, sample_base, emit_outgoing, n_local;   // n_local: pixels per chunk (path id = slot * n_local + pixel in chunk)
    uint32_t pix_base;                                                  // first local pixel of the chunk
    uint32_t partition;      // RT_OPT_SHADE_PARTITION: bit 0 = hits first / misses last inside every block, bit 1 = each block's
                             // outgoing and shadow rays grouped by direction octant
    uint32_t final_bounce;   // 1: no shade launch follows for these paths (bounce == max_bounces)
    uint32_t count_in_ray;   // 1 (rt_integrate): a path's number of log entries travels with its ray (thr.w) and cnt[id] is
                             // written once, when the path ends; 0 (stage API): cnt[id] is read and written at every bounce,
                             // so that the radiance can be read between any two stages
};

// SampleBlueNoise, sampling.h:40-61 (Heitz et al. 2019 tables, values 0..255).  The reference
// indexes rankingTile with the un-wrapped dimension (sampling.h:50, no `% 8`), which runs past
// the end of the table for the last pixels of a tile row; entries past the end read as 0 here
// (oracle and reference-kernel build pad the table the same way).
RT_DEV float SampleBlueNoise(const ShadeArgs& a, uint32_t px, uint32_t py, uint32_t sample_index, uint32_t dim)
{
    int pixel_i = (int)px & 127, pixel_j = (int)py & 127;
    int sampleIndex = (int)sample_index & 255, sampleDimension = (int)dim & 255;
    int ridx = sampleDimension + (pixel_i + pixel_j * 128) * 8;
    int rank = ridx < 128 * 128 * 8 ? (int)a.bn_rank[ridx] : 0;
    int rankedSampleIndex = sampleIndex ^ rank;
    int value = (int)a.bn_sobol[sampleDimension + rankedSampleIndex * 256];
    value = value ^ (int)a.bn_scramble[(sampleDimension % 8) + (pixel_i + pixel_j * 128) * 8];
    return (0.5f + (float)value) / 256.0f;
}

// One queue entry of k_shade -- Miss (miss.cl:65-76) or HitSurface (hit_surface.cl:79-184) with the log entries, the deferred direct sample and the
// BSDF sample -- as a function of its own: k_shade calls it per thread and compacts the outputs per block, k_frame (frame_kernels.h) per lane of a
// wave that carries its own paths through the bounces.
template <bool FURNACE, bool BLUE, bool NEE, bool COMPACT>
RT_DEV void shade_entry(const DScene& sc, const DTile& tile, const ShadeArgs& a, const uint32_t i, bool& want_shadow, bool& want_next, bool& no_block,
    uint32_t& next_flag, float4& sh_o, float4& sh_d, float4& nx_o, float4& nx_d, float4& nx_t, uint32_t& sh_entry)
{
    float4 hit = a.hits[i];
    float4 rd = a.in_d4[i];
    uint32_t prim = __float_as_uint(hit.z);
    uint32_t id = __float_as_uint(rd.w);                               // slot * n_local + local pixel
    uint32_t slot = id / a.n_local;
    uint32_t pix = id - slot * a.n_local;
    uint32_t sample_idx = a.sample_base + slot;
    float4 thr4 = a.in_thr[i];
    uint32_t nlog = a.count_in_ray ? (__float_as_uint(thr4.w) & 0x7FFFFFFFu) : a.log.cnt[id];   // contributions logged so far
    // compact log: this bounce may write entries nlog and nlog + 1; those beyond the inline rows go to the path's overflow
    // block, which the previous bounce allocated (below) whenever that could happen
    uint32_t oblk = RT_EMPTY_REF;
    if (COMPACT && nlog + 1u >= a.log.inline_entries) oblk = a.log.ovf_slot[id];
    no_block = oblk == RT_EMPTY_REF;
    // one entry of this path's log (the full layout: row nlog, column id)
    auto put = [&](float x, float y, float z)
    {
        if (COMPACT) log_put(a.log, nlog, id, oblk, x, y, z);
        else log_store(a.log.rlog, (size_t)nlog * a.log.stride + id, x, y, z);
    };
    // NEE: bit 31 of the same word = the path's last scattering event was a delta one (set by the previous bounce)
    const bool prev_delta = NEE && (__float_as_uint(thr4.w) >> 31) != 0u;
    f3 hit_throughput = F3(thr4.x, thr4.y, thr4.z);

    if (prim == RT_INVALID_ID)
    {
        // Miss, miss.cl:65-76
        f3 sky = FURNACE ? F3s(0.5f) : SampleSky(sc, F3(rd.x, rd.y, rd.z));
        f3 add = sky * hit_throughput;
        put(add.x, add.y, add.z);   // radiance[pix] += ...
        ++nlog;
    }
    else
    {
        // HitSurface, hit_surface.cl:79-184
        f3 incoming = F3(-rd.x, -rd.y, -rd.z);
        uint32_t lp = a.pix_base + pix;                                  // local pixel of the tile
        uint32_t ly = lp / tile.width;
        uint32_t px = lp - ly * tile.width;
        uint32_t py = tile_global_row(tile, ly);

        const float4* tp = sc.tris_sh + (size_t)prim * 8;
        float4 q0 = tp[0], q1 = tp[1], q2 = tp[2], q3 = tp[3], q4 = tp[4], q5 = tp[5], q6 = tp[6];
        f3 p1 = xyz(q0), p2 = xyz(q1), p3 = xyz(q2);
        f3 n1 = xyz(q3), n2 = xyz(q4), n3 = xyz(q5);
        float bu = hit.x, bv = hit.y;
        float w0 = 1.0f - bu - bv;
        f3 position = p1 * w0 + p2 * bu + p3 * bv;
        f3 geometry_normal = normalize3(cross3(p2 - p1, p3 - p1));
        f2 texcoord;
        texcoord.x = q0.w * w0 + q2.w * bu + q4.w * bv;                // uv1.x, uv2.x, uv3.x
        texcoord.y = q1.w * w0 + q3.w * bu + q5.w * bv;                // uv1.y, uv2.y, uv3.y
        f3 normal = normalize3(n1 * w0 + n2 * bu + n3 * bv);

        Material material;
        ApplyTextures(sc, __float_as_uint(q6.x), material, texcoord);

        // NEE: light gathered from the emissive triangles by next-event estimation is not counted again when a
        // scattered ray happens to hit one -- emission is added for camera rays and after delta events only
        if (!FURNACE && (!NEE || a.bounce == 0u || prev_delta))
        {
            if (material.emission.x * 1.0f + material.emission.y * 1.0f + material.emission.z * 1.0f > 0.0f)
            {
                f3 e = hit_throughput * material.emission;
                put(e.x, e.y, e.z);         // radiance[pix] += ...
                ++nlog;
            }
        }

        uint32_t sample_seed = BLUE ? 0u : SampleRandomSampleSeed(SampleRandomPixelSeed(px, py), sample_idx);
        // SampleRandom(x, y, sample, bounce, type), sampling.h:64-82
        auto draw = [&](uint32_t type) -> float
        {
            return BLUE ? SampleBlueNoise(a, px, py, sample_idx, a.bounce * 5u + type)
                        : SampleRandomDim(sample_seed, a.bounce, type);
        };

        // Direct lighting :115-145 (Light_Sample light.h:30-65)
        {
            float s_light = draw(4);
            const uint32_t n_lights = NEE ? sc.light_count + sc.emissive_count : sc.light_count;
            int light_idx = cl_clampi((int)(s_light * (float)n_lights), 0, (int)n_lights - 1);
            float pdf = 1.0f / (float)n_lights;
            f3 light_radiance;
            f3 outgoing;
            if (!NEE || (uint32_t)light_idx < sc.light_count)
            {
                float4 lo = sc.lights[light_idx * 3 + 0], lr = sc.lights[light_idx * 3 + 1];
                uint32_t ltype = __float_as_uint(sc.lights[light_idx * 3 + 2].x);
                light_radiance = xyz(lr);
                if (ltype == RT_LIGHT_TYPE_POINT)
                {
                    f3 to_light = xyz(lo) - position;
                    float sq_length = dot3(to_light, to_light);
                    light_radiance = light_radiance / sq_length;
                    outgoing = to_light;
                }
                else
                {
                    outgoing = xyz(lo) * RT_MAX_RENDER_DIST;
                }
            }
            else
            {
                // an emissive triangle, sampled uniformly by area: u1 = what the index left of s * n, u2 = the
                // BSDF-layer sample of this bounce (oracle.c: Light_SampleWithEmissive, the same operations)
                const uint32_t lt = sc.emissive[(uint32_t)light_idx - sc.light_count];
                float u1 = s_light * (float)n_lights - (float)light_idx;
                u1 = cl_min(cl_max(u1, 0.0f), 1.0f);
                const float su = __builtin_sqrtf(u1);
                const float b0 = 1.0f - su, b1 = draw(1) * su;
                const float b2 = 1.0f - b0 - b1;
                const float4* lq = sc.tris_sh + (size_t)lt * 8;
                const float4 l0 = lq[0], l1 = lq[1], l2 = lq[2], l3 = lq[3], l4 = lq[4], l5 = lq[5], l6 = lq[6];
                const f3 a1 = xyz(l0), a2 = xyz(l1), a3 = xyz(l2);
                const f3 lp = a1 * b0 + a2 * b1 + a3 * b2;
                f2 luv;
                luv.x = l0.w * b0 + l2.w * b1 + l4.w * b2;
                luv.y = l1.w * b0 + l3.w * b1 + l5.w * b2;
                Material lm;
                ApplyTextures(sc, __float_as_uint(l6.x), lm, luv);
                const f3 nl = cross3(a2 - a1, a3 - a1);                  // length = 2 * area
                const f3 to_light = lp - position;
                const float d2 = dot3(to_light, to_light);
                float g = 0.0f;
                // front side only: the reference's ray-triangle test culls back faces (det = -dir . nl < 1e-8), so a
                // triangle is visible, and its emission counted, only from the side its normal points to
                const float dist = __builtin_sqrtf(d2);
                const float nd = -dot3(nl, to_light);                    // 2 * area * d * cos_l
                if (d2 > 0.0f && nd / dist >= 1e-8f) g = (nd * 0.5f) / (dist * d2);   // cos_l * area / d^2
                // the shadow ray starts EPS along the normal but is aimed from `position`: it meets the emitter's plane
                // up to EPS / |cos_l| early.  Stop short by twice that + 2^-10 d; drop what leaves nothing (oracle.c)
                float keep = 1.0f - 0.0009765625f - (2.0f * RT_EPS) * __builtin_sqrtf(dot3(nl, nl)) / nd;
                if (!(keep > 0.0f) || !(g > 0.0f)) { keep = 1.0f; g = 0.0f; }
                outgoing = to_light * keep;
                light_radiance = lm.emission * g;
            }
            float distance_to_light = length3(outgoing);
            outgoing = normalize3(outgoing);
            f3 brdf = EvaluateMaterial(material, normal, incoming, outgoing);
            float m = cl_max(dot3(outgoing, normal), 0.0f);
            f3 lsamp = ((light_radiance * hit_throughput) * brdf / pdf) * m;
            want_shadow = (pdf > 0.0f) && (dot3(lsamp, lsamp) > 0.0f);
            f3 so = position + normal * RT_EPS;
            sh_o = make_float4(so.x, so.y, so.z, distance_to_light);
            sh_d = make_float4(outgoing.x, outgoing.y, outgoing.z, __uint_as_float(id));
            sh_entry = nlog;
            if (want_shadow)
            {
                // deferred direct sample (direct_light_samples_buffer_): logged now, retracted
                // by the shadow trace if the light turns out to be occluded
                put(lsamp.x, lsamp.y, lsamp.z);
                ++nlog;
            }
        }

        // Indirect lighting :148-184.  On the last bounce the outgoing ray is never traced (and its throughput
        // never read): the whole block is skipped, wave-uniformly.
        if (a.emit_outgoing != 0)
        {
            f2 s;
            s.x = draw(2);
            s.y = draw(3);
            float s1 = draw(1);
            float pdf = 0.0f;
            f3 outgoing;
            float offset;
            bool delta;
            f3 bxdf = SampleBxdf<FURNACE>(s1, s, material, normal, incoming, outgoing, pdf, offset, delta);
            if (NEE && delta) next_flag = 0x80000000u;
            f3 throughput = F3s(0.0f);
            if ((double)pdf > 0.0) throughput = bxdf / pdf;
            f3 new_thr = hit_throughput * throughput;                 // throughputs[pixel] *= throughput
            want_next = (double)pdf > 0.0;
            f3 oo = position + geometry_normal * RT_EPS * offset;
            nx_o = make_float4(oo.x, oo.y, oo.z, RT_MAX_RENDER_DIST);
            nx_d = make_float4(outgoing.x, outgoing.y, outgoing.z, rd.w);
            nx_t = make_float4(new_thr.x, new_thr.y, new_thr.z, 0.0f);
        }
    }
    nx_t.w = __uint_as_float(nlog | next_flag);
    if (!a.count_in_ray || ((!want_next || a.final_bounce) && nlog != 0u)) a.log.cnt[id] = nlog;   // the path's final count is what k_flush replays
}

// NEE: the scene asked for next-event estimation over the emissive triangles too (RT_SCENE_EMISSIVE_NEE, an opt-in
// extension: DESIGN.md 7b); the other instances are the reference's estimator.
// COMPACT: the radiance log is in its compact layout (DLog; RT_OPT_COMPACT_LOG): separate instances, so that the default ones
// carry none of its code.
template <bool FURNACE, bool BLUE, bool NEE = false, bool COMPACT = false>
__global__ __launch_bounds__(RT_SHADE_BLOCK) __attribute__((amdgpu_waves_per_eu(NEE ? 5 : RT_SHADE_WAVES))) void k_shade(DScene sc, DTile tile, ShadeArgs a)
{
    const uint32_t count = a.counters->queue[a.bounce];
    // the closest-hit trace of this bounce has completed (stream order): rewind the
    // work heads for the shadow trace of this bounce and the closest trace of the next
    {
        const uint32_t g = blockIdx.x * RT_SHADE_BLOCK + threadIdx.x;
        // the work heads of the two launches this bounce feeds: closest of bounce + 1, shadow of this bounce (flavour
        // 1 + (bounce & 1); the other shadow flavour may still be in use by the previous bounce's shadow trace)
        if (g < 16)
        {
            const uint32_t fl = g < 8 ? 0u : 1u + (a.bounce & 1u);
            a.counters->head[fl][g & 7] = 0; a.counters->slow_head[fl][g & 7] = 0;
            if ((g & 7) == 0) a.counters->slow_count[fl] = 0;
        }
    }
    if (blockIdx.x * RT_SHADE_BLOCK >= count) return;                    // whole block idle (uniform)
    // Hits first, misses last inside the block's 512 queue entries: a wave then holds (almost) only hits or only
    // misses, and the two divergent halves of the kernel -- SampleSky's fp64 atan2 / acos for an escaped ray,
    // material + light + BSDF code for a hit -- are no longer both executed by every wave.  (Round 1 tried this when
    // the kernel was bound by its scattered counter traffic: no gain; with that gone the vector ALU is its busiest
    // unit and the partition pays.)  Any order of the entries gives the same image: everything is keyed by path id.
    uint32_t i = blockIdx.x * RT_SHADE_BLOCK + threadIdx.x;
    if (a.partition)
    {
        __shared__ uint16_t s_perm[RT_SHADE_BLOCK];
        __shared__ uint32_t s_hit[RT_SHADE_BLOCK / 64], s_miss[RT_SHADE_BLOCK / 64];
        const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
        const bool valid = i < count;
        const bool is_miss = valid && __float_as_uint(a.hits[valid ? i : 0u].z) == RT_INVALID_ID;
        const unsigned long long mh = __ballot(valid && !is_miss), mm = __ballot(is_miss);
        if (lane == 0) { s_hit[wave] = (uint32_t)__popcll(mh); s_miss[wave] = (uint32_t)__popcll(mm); }
        __syncthreads();
        uint32_t hits_before = 0, miss_before = 0, hits_total = 0;
        for (uint32_t w = 0; w < RT_SHADE_BLOCK / 64; ++w)
        {
            hits_total += s_hit[w];
            if (w < wave) { hits_before += s_hit[w]; miss_before += s_miss[w]; }
        }
        const unsigned long long lt = (1ull << lane) - 1ull;
        if (valid)
            s_perm[is_miss ? hits_total + miss_before + (uint32_t)__popcll(mm & lt) : hits_before + (uint32_t)__popcll(mh & lt)] =
                (uint16_t)threadIdx.x;
        __syncthreads();
        if (valid) i = blockIdx.x * RT_SHADE_BLOCK + s_perm[threadIdx.x];   // valid entries are a prefix of the block
    }
    const bool active = i < count;

    bool want_shadow = false, want_next = false;
    bool no_block = false;             // compact log: this path has no overflow block yet
    uint32_t next_flag = 0;            // NEE: bit 31 of the outgoing ray's log-count word = "this event was a delta one"
    float4 sh_o = make_float4(0, 0, 0, 0), sh_d = sh_o, nx_o = sh_o, nx_d = sh_o, nx_t = sh_o;
    uint32_t sh_entry = 0;

    if (active) shade_entry<FURNACE, BLUE, NEE, COMPACT>(sc, tile, a, i, want_shadow, want_next, no_block, next_flag, sh_o, sh_d, nx_o, nx_d, nx_t, sh_entry);
    // Compact log: the next bounce writes entries nlog, nlog + 1 -- a path that goes on and could cross the inline rows gets its
    // overflow block now (nx_t.w = its entries so far, nx_d.w = its id).  Entries so far <= 2 (bounce + 1): before that reaches
    // inline_entries - 1 no path of the launch can want one.
    if (COMPACT && 2u * (a.bounce + 1u) + 1u >= a.log.inline_entries && !a.final_bounce)
    {
        const bool want_block = want_next && no_block && (__float_as_uint(nx_t.w) & 0x7FFFFFFFu) + 1u >= a.log.inline_entries;
        // Per WAVE, and only in waves where somebody wants one (a tenth of the paths ever do): one atomic on one of
        // RT_LOG_SUBPOOLS bump counters, each owning an equal share of the pool -- no block-wide barrier on k_shade's critical
        // path (round 3 allocated per 512-thread block: two __syncthreads and 0.024 ms per sample, which is why the compact
        // layout was not the default).  A share that runs dry raises the flag like the whole pool did.
        unsigned long long wm = __ballot(want_block);
        if (wm != 0ull)
        {
            const uint32_t lane = threadIdx.x & 63u;
            const uint32_t home = blockIdx.x * (RT_SHADE_BLOCK / 64u) + (threadIdx.x >> 6);
            // Fewer sub-pools for a small pool (at least 32 blocks each: a tile of a few hundred paths used to get shares of ONE block, a
            // dry share at once and a fallback to the full layout for good -- ADVICE r04); the division's remainder belongs to the last share.
            uint32_t n_pools = RT_LOG_SUBPOOLS;
            while (n_pools > 1u && a.log.ovf_blocks / n_pools < 32u) n_pools >>= 1;
            const uint32_t share = a.log.ovf_blocks / n_pools, last_share = a.log.ovf_blocks - share * (n_pools - 1u);
            uint32_t got = RT_EMPTY_REF;
            // the wave's own share first, then up to seven others (a share that is full does not end the batch while its
            // neighbours have room: small launches use few shares, late bounces use them unevenly)
            for (uint32_t attempt = 0; attempt < (n_pools < 8u ? n_pools : 8u) && wm != 0ull; ++attempt)
            {
                const uint32_t pool = (home + attempt * (n_pools > 8u ? 9u : 1u)) & (n_pools - 1u);
                uint32_t base = 0;
                const int first = __builtin_ctzll(wm);
                if ((int)lane == first) base = atomicAdd(&a.counters->log_ovf_next[pool], (uint32_t)__popcll(wm));
                base = (uint32_t)__shfl((int)base, first, 64);
                if (want_block && got == RT_EMPTY_REF)
                {
                    const uint32_t k = base + (uint32_t)__popcll(wm & ((1ull << lane) - 1ull));
                    if (k < (pool == n_pools - 1u ? last_share : share)) got = pool * share + k;
                }
                wm = __ballot(want_block && got == RT_EMPTY_REF);
            }
            if (want_block)
            {
                if (got == RT_EMPTY_REF) a.counters->log_ovf_flag = 1u;      // dry: the host repeats this batch in the full layout
                a.log.ovf_slot[__float_as_uint(nx_d.w)] = got;
            }
        }
    }

    uint32_t sidx, nidx;
    if (a.partition & 2u)
    {
        // (16 groups -- the octant, then whether the direction leans to x or to y -- trace no faster and cost k_shade more)
        const uint32_t key_s = (sh_d.x < 0.0f ? 1u : 0u) | (sh_d.y < 0.0f ? 2u : 0u) | (sh_d.z < 0.0f ? 4u : 0u);
        const uint32_t key_n = (nx_d.x < 0.0f ? 1u : 0u) | (nx_d.y < 0.0f ? 2u : 0u) | (nx_d.z < 0.0f ? 4u : 0u);
        block_append2_keyed<8>(want_shadow, key_s, want_next, key_n, &a.counters->shadow[a.bounce], &a.counters->queue[a.bounce + 1], sidx, nidx);
    }
    else
        block_append2(want_shadow, want_next, &a.counters->shadow[a.bounce], &a.counters->queue[a.bounce + 1], sidx, nidx);
    if (want_shadow)
    {
        a.sh_o4[sidx] = sh_o;
        a.sh_d4[sidx] = sh_d;
        a.sh_aux[sidx] = sh_entry;
    }
    if (want_next)
    {
        a.out_o4[nidx] = nx_o;
        a.out_d4[nidx] = nx_d;
        a.out_thr[nidx] = nx_t;
    }
}

// Replays the radiance log: for every pixel, sample slot by sample slot, contribution
// by contribution -- the exact order in which the reference's kernels executed
// `radiance[pixel] += ...` (miss.cl:75, hit_surface.cl:110, accumulate_direct_samples.cl:51).
// keep_open (a mid-sample read through the stage API on a COMPACT allocation): the sample goes on, and its count and its
// overflow block must stay what k_shade knows them to be -- the entries replayed here are zeroed instead (adding +0.0 again
// at the sample's end is the identity), the count is kept.  The full layout restarts the count at 0 as it always did.
// first_slot (RT_OPT_SAMPLES_AHEAD): the replay covers the sample slots first_slot .. first_slot + n_slots - 1 only -- a batch traced ahead of the
// caller's Integrate() calls reaches the radiance one sample per call, in sample order.
__global__ __launch_bounds__(256) void k_flush(float4* __restrict__ radiance, DLog log, uint32_t n_pixels, uint32_t n_slots, uint32_t id_stride,
    uint32_t keep_open, uint32_t first_slot)
{
    // radiance: already offset to the chunk's first pixel; id_stride: pixels per chunk as allocated
    uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n_pixels) return;
    float4 r = radiance[p];
    for (uint32_t slot = first_slot; slot < first_slot + n_slots; ++slot)
    {
        uint32_t id = slot * id_stride + p;
        uint32_t c = log.cnt[id];
        const uint32_t oblk = c > log.inline_entries ? log.ovf_slot[id] : 0u;       // compact layout: the path's overflow block
        for (uint32_t k = 0; k < c; ++k)
        {
            const size_t at = log_index(log, k, id, oblk);
            if (at == ~(size_t)0) continue;                                          // no home (pool ran dry: log_put skipped it too)
            const rt_rgb v = *reinterpret_cast<const rt_rgb*>(log.rlog + 3 * at);
            r.x += v.x; r.y += v.y; r.z += v.z;
            if (keep_open) log_store(log.rlog, at, 0.0f, 0.0f, 0.0f);
        }
        if (c && !keep_open) log.cnt[id] = 0;
    }
    radiance[p] = r;
}
