// raygen_kernels.h -- sample begin + ray generation, counter folding, debug evaluation of the
// device math (raygeneration.cl, clear_counter.cl / increment_counter.cl).
#pragma once
#include "kernels_common.h"

// The primary ray of global pixel (pixel_x, pixel_y) for sample `sample_idx` (raygeneration.cl:92-133): origin on the lens, direction through the
// focus plane.  Shared by k_raygen and k_frame (frame_kernels.h).
RT_DEV void raygen_ray(const DTile& tile, const rt_camera& cam, float tan_half_fov, uint32_t pixel_x, uint32_t pixel_y, uint32_t sample_idx,
    f3& new_pos, f3& d)
{
    uint32_t pixel_idx = pixel_y * tile.width + pixel_x;                 // GLOBAL pixel index
    float inv_width = 1.0f / (float)tile.width;
    float inv_height = 1.0f / (float)tile.height;
    uint32_t seed = pixel_idx + (1103515245u * sample_idx + 12345u);     // :61,98

    float x = ((float)pixel_x + GetRandomFloat(seed)) * inv_width;
    float y = ((float)pixel_y + GetRandomFloat(seed)) * inv_height;

    float angle = tan_half_fov;                                          // rt_tanf(0.5f * fov), host-evaluated
    x = (x * 2.0f - 1.0f) * angle * cam.aspect_ratio;
    y = (y * 2.0f - 1.0f) * angle;

    f3 front = F3(cam.front.x, cam.front.y, cam.front.z);
    f3 up = F3(cam.up.x, cam.up.y, cam.up.z);
    f3 pos = F3(cam.position.x, cam.position.y, cam.position.z);
    f3 right = cross3(front, up);
    f3 dir = normalize3(right * x + up * y + front);

    f3 point_aimed = pos + dir * cam.focus_distance;
    // PointInHexagon :40-49 (index 3 = the reference's out-of-bounds read, defined as (0,0))
    int hidx = (int)__builtin_floorf(GetRandomFloat(seed) * 3.0f);
    int h1 = hidx > 3 ? 3 : hidx;
    int h2 = (hidx + 1) % 3;
    float hx1 = h1 == 0 ? -1.0f : (h1 == 3 ? 0.0f : 0.5f);
    float hy1 = h1 == 1 ? 0.866f : (h1 == 2 ? -0.866f : 0.0f);
    float hx2 = h2 == 0 ? -1.0f : 0.5f;
    float hy2 = h2 == 1 ? 0.866f : (h2 == 2 ? -0.866f : 0.0f);
    float p1 = GetRandomFloat(seed);
    float p2 = GetRandomFloat(seed);
    float dofx = p1 * hx1 + p2 * hx2;
    float dofy = p1 * hy1 + p2 * hy2;
    float r = cam.aperture;
    new_pos = pos + right * (dofx * r) + up * (dofy * r);
    d = normalize3(point_aimed - new_pos);

}

// ---------------------------------------------------------------------------
// sample begin + ray generation (raygeneration.cl:65-139)
// ---------------------------------------------------------------------------
// The tile may be rendered in CHUNKS of pixels (RT_OPT_PATH_STATE_LIMIT_MB): this launch covers local pixels
// chunk_base .. chunk_base + chunk_count - 1; path ids are chunk-relative (slot * chunk_stride + pixel in chunk).
__global__ __launch_bounds__(256) void k_raygen(DTile tile, rt_camera cam, uint32_t sample_base, uint32_t n_slots,
    float tan_half_fov, uint32_t prev_bounces, float4* __restrict__ o4, float4* __restrict__ d4,
    float4* __restrict__ thr, DCounters* __restrict__ counters, uint32_t chunk_base,
    uint32_t chunk_count, uint32_t chunk_stride, uint32_t prev_accumulates)
{
    uint32_t n_total = chunk_count * n_slots;                            // n_slots samples in flight
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i == 0)
    {
        // fold the previous sample's per-bounce counters into the totals, then
        // clear them (replaces the ClearCounter launches, cl_pt_integrator.cpp:651-663)
        unsigned long long c = 0, s = 0;
        for (uint32_t b = 0; b <= prev_bounces && b < 64; ++b)
        {
            c += counters->queue[b];
            s += counters->shadow[b];
            // last_*: per-bounce counts of the most recent BATCH: the chunks of one batch add up
            counters->last_queue[b] = (prev_accumulates ? counters->last_queue[b] : 0u) + counters->queue[b];
            counters->last_shadow[b] = (prev_accumulates ? counters->last_shadow[b] : 0u) + counters->shadow[b];
        }
        counters->total_closest += c;
        counters->total_shadow += s;
        for (uint32_t b = 0; b < 64; ++b) { counters->queue[b] = 0; counters->shadow[b] = 0; }
        counters->queue[0] = n_total;                                    // raygeneration.cl:135-138
    }
    if (i < RT_LOG_SUBPOOLS) counters->log_ovf_next[i] = 0;              // a new sequence: its log's overflow pool is empty
    if (i < 24) counters->head[i >> 3][i & 7] = 0;                       // the next trace launches start from 0
    if (i >= n_total) return;

    // queue order is PIXEL-major: the n_slots samples of a pixel sit next to each other, so
    // a wave holds a few pixels x all their samples -- nearly identical primary rays, and
    // secondary/shadow rays that start from the same small surface patch (shadow rays towards
    // a directional light are then almost parallel AND co-located).  Fewer distinct BVH
    // records per load instruction is what the L1 data path rewards.  The path id keeps the
    // slot-major form (slot * n_local + pixel) the radiance log is laid out by.  (Pixel-major ids were tried in
    // round 2: k_shade's log writes and the shadow kernel's retractions then share cache lines, -5 % / -2 % on those
    // kernels, but k_flush -- one thread per pixel walking its samples in order -- loses its coalescing and goes from
    // 3.7 to 20 ms per batch: a net loss.)
    uint32_t cp = i / n_slots;                                           // pixel of this chunk
    uint32_t slot = i - cp * n_slots;
    uint32_t lp = chunk_base + cp;                                       // local pixel of this tile
    uint32_t sample_idx = sample_base + slot;
    uint32_t ly = lp / tile.width;
    uint32_t pixel_x = lp - ly * tile.width;
    uint32_t pixel_y = tile_global_row(tile, ly);

    f3 new_pos, d;
    raygen_ray(tile, cam, tan_half_fov, pixel_x, pixel_y, sample_idx, new_pos, d);
    o4[i] = make_float4(new_pos.x, new_pos.y, new_pos.z, RT_MAX_RENDER_DIST);
    d4[i] = make_float4(d.x, d.y, d.z, __uint_as_float(slot * chunk_stride + cp));   // path id
    thr[i] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
}


// end-of-run fold of the per-bounce counters (same as the prologue of k_raygen)
__global__ void k_fold_counters(DCounters* counters, uint32_t bounces, uint32_t prev_accumulates)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned long long c = 0, s = 0;
    for (uint32_t b = 0; b <= bounces && b < 64; ++b)
    {
        c += counters->queue[b];
        s += counters->shadow[b];
        counters->last_queue[b] = (prev_accumulates ? counters->last_queue[b] : 0u) + counters->queue[b];
        counters->last_shadow[b] = (prev_accumulates ? counters->last_shadow[b] : 0u) + counters->shadow[b];
        counters->queue[b] = 0;
        counters->shadow[b] = 0;
    }
    counters->total_closest += c;
    counters->total_shadow += s;
}


// device-math known-answer hook (rt_debug_eval)
__global__ void k_debug_eval(int fn, const float* a, const float* b, float* out, uint32_t n)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = a[i], y = b ? b[i] : 0.0f, r = 0.0f;
    switch (fn)
    {
    case 0: r = rt_sinf(x); break;
    case 1: r = rt_cosf(x); break;
    case 2: r = rt_tanf(x); break;
    case 3: r = rt_powf(x, y); break;
    case 4: r = rt_atan2f(x, y); break;
    case 5: r = rt_acosf(x); break;
    case 6: r = __builtin_sqrtf(x); break;
    case 7: r = x / y; break;
    case 8:
    {
        uint32_t px = __float_as_uint(x) & 0xFFFFu, py = __float_as_uint(x) >> 16;
        uint32_t smp = __float_as_uint(y) & 0xFFFFu, dim = __float_as_uint(y) >> 16;
        uint32_t ss = SampleRandomSampleSeed(SampleRandomPixelSeed(px, py), smp);
        r = SampleRandomDim(ss, dim / 5u, dim % 5u);
        break;
    }
    case 9:   // fp64 path of GGX_Sample: x = alpha*alpha*s.y, y = s.y
        r = (float)(1.0 / __builtin_sqrt(1.0 + (double)x / (1.0 - (double)y)));
        break;
    default: break;
    }
    out[i] = r;
}
