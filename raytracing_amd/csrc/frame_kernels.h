// frame_kernels.h -- k_frame: ONE launch per frame of the reference's frame-by-frame pattern (Render::RenderFrame -> Integrator::Integrate,
// src/render.cpp:197, src/integrator/integrator.cpp:27-59: one sample per pixel, bounce after bounce), opt-in (RT_OPT_FRAME_KERNEL).
//
// The stage kernels pay that pattern 3 x (B + 1) launch boundaries: every launch lasts as long as its longest ray while the machine drains, and the
// next one cannot start before.  Here a WAVE carries its own pixels -- a few chunks of 64, interleaved over its XCD's eighth of the tile -- through
// the whole sample: primary rays, then per bounce the closest-hit walk of its rays (w4_trace_body<PRIVATE>: the body of k_trace_w4 over a queue of
// the wave's own, idle lanes refilled in order, no atomics), shade_entry per lane with the outgoing and the shadow rays compacted by ballot into
// the wave's own chunks of the next queues, the shadow walk, and at the end the replay of its pixels' radiance log.  No launch boundary, no
// grid-wide barrier, no shared counter: a wave that is draining its last rays of a bounce shares its SIMD with waves that are anywhere else in
// their frames.  Per path everything is what the stage kernels do -- the same functions, the same order of log entries -- so the radiance is the
// reference's bit for bit; only queue ORDER differs (results are keyed by path id).  Ray counters: per-wave rows, summed by k_frame_sum.
#pragma once
#include "raygen_kernels.h"
#include "trace_kernels.h"
#include "shade_kernels.h"

#define RT_FRAME_COUNT_STRIDE 136u       // per-wave row: [0..63] closest rays per bounce, [64..127] shadow rays per bounce, [128] spills, [129] slow rays, [130] scratch,
                                         // [131..134] the wave's 100 MHz ticks in the closest walks / shading / shadow walks / in all (rt_frame_debug_frame_rows)

struct FrameArgs
{
    ShadeArgs shade;                     // log, sampler tables, n_local, sample_base ...; the queue pointers are set per bounce from the arrays below
    float4* o4[2]; float4* d4[2]; float4* thr[2];      // ping-pong ray queues (a wave uses ITS chunks of them)
    float4* hits;
    float4* sh_o4; float4* sh_d4; uint32_t* sh_aux;    // shadow queue (traced by the same wave right after it is filled)
    float4* radiance;
    uint2* spill;                        // per-lane stack spill area: (blocks * 64) x (RT_W4_STACK_MAX - 12) entries
    uint32_t* slow_list;                 // per-wave list of queue indices the wide walk left out: chunks_per_wave * 64 entries each
    uint32_t* wave_counts;               // blocks x RT_FRAME_COUNT_STRIDE
    rt_camera cam;
    float tan_half_fov;
    uint32_t max_bounces, drop_last, tune, tail_q, chunks_per_wave;
};

#ifndef RT_FRAME_WAVES             // waves per SIMD k_frame is compiled for (tools/build_variants.py)
#define RT_FRAME_WAVES 5          // 96 VGPRs + 27 dwords of scratch; at 4 (126 VGPRs, nothing spilled) 3 - 5 % slower on every config (profiles/r05_call13.log)
#endif
#ifndef RT_FRAME_XCD_REGIONS       // 1: a wave's chunks come from its XCD's eighth of the tile (the stage kernels' rule: one image region per L2);
#define RT_FRAME_XCD_REGIONS 0     // 0: from the whole tile -- pixels are owned for the WHOLE frame here, and an XCD whose eighth is sky idles (below)
#endif
template <bool FURNACE, bool BLUE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RT_FRAME_WAVES, 6))) void k_frame(DScene sc, DTile tile, FrameArgs fa)
{
    __shared__ uint2 stack[12][64];
    const uint32_t lane = threadIdx.x, w = blockIdx.x, G = gridDim.x;
    const uint32_t n_local = fa.shade.n_local;
    const uint32_t n_chunks = (n_local + 63u) >> 6;
    // This wave's chunks: w, w + G, w + 2 G ... of the tile's chunks -- a sample of the WHOLE image.  (With the stage kernels' rule -- slot, slot + S ...
    // of the XCD's contiguous eighth of the queue -- the city block's frame took 4.36 ms against 3.37 with the stage kernels while the closed Cornell box
    // took 2.50 against 3.99: a queue is compacted per launch, so its eighths are balanced, but pixels are owned for the whole frame here, and the XCDs
    // whose eighth of the image is sky ran dry after the first bounce.  profiles/r05_call12.log)
#if RT_FRAME_XCD_REGIONS
    const uint32_t xcd = w & 7u, slot = w >> 3, S = G >> 3;
    const uint32_t cpx = (n_chunks + 7u) >> 3;
    const uint32_t first = xcd * cpx + slot;
    uint32_t my_chunks = 0;
    while (slot + my_chunks * S < cpx && first + my_chunks * S < n_chunks) ++my_chunks;
#else
    const uint32_t first = w, S = G;
    const uint32_t my_chunks = w < n_chunks ? (n_chunks - 1u - w) / G + 1u : 0u;
#endif
    uint32_t* const row = fa.wave_counts + (size_t)w * RT_FRAME_COUNT_STRIDE;
    if (lane == 0) { row[128] = 0; row[129] = 0; }
    for (uint32_t k = lane; k < 128u; k += 64u) row[k] = 0;
    if (my_chunks == 0) return;
    auto index_of = [&](uint32_t p) { return (first + (p >> 6) * S) * 64u + (p & 63u); };     // position p of the wave's queue -> queue index
    const uint32_t last_base = (first + (my_chunks - 1u) * S) * 64u;
    const uint32_t n_pixels = (my_chunks - 1u) * 64u + (n_local - last_base < 64u ? n_local - last_base : 64u);   // the tile's last chunk may be partial
    uint32_t* const my_slow = fa.slow_list + (size_t)w * fa.chunks_per_wave * 64u;
    uint2* const lane_spill = fa.spill + (size_t)(w * 64u + lane) * (RT_W4_STACK_MAX - 12);
    auto spill_push = [&](int sp, uint2 e) { lane_spill[sp] = e; };
    auto spill_pop = [&](int sp) -> uint2 { return spill_load64(lane_spill + sp); };
    DLog log = fa.shade.log;

    // ---- primary rays (k_raygen's per-ray part; path id = local pixel: one sample in flight, the whole tile one chunk) ----
    for (uint32_t p = lane; p < n_pixels; p += 64u)
    {
        const uint32_t lp = index_of(p);
        const uint32_t ly = lp / tile.width, px = lp - ly * tile.width, py = tile_global_row(tile, ly);
        f3 o, d;
        raygen_ray(tile, fa.cam, fa.tan_half_fov, px, py, fa.shade.sample_base, o, d);
        fa.o4[0][lp] = make_float4(o.x, o.y, o.z, RT_MAX_RENDER_DIST);
        fa.d4[0][lp] = make_float4(d.x, d.y, d.z, __uint_as_float(lp));
        fa.thr[0][lp] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
    }
    // What the wave's lanes wrote is read by OTHER lanes of the same wave next: a workgroup-scope fence (the stores have left the wave; the CU's L1
    // is written through and serves its own CU's later loads).  NOT __threadfence(): at agent scope that is a write-back of the XCD's whole L2 and an
    // invalidation of the L1 -- 36 of them per wave and frame made the first version of this kernel 2.5 x SLOWER than the stage kernels
    // (8.57 against 3.37 ms per frame, profiles/r05_call12.log).
    __threadfence_block();

    uint32_t n = n_pixels, cur = 0;
    const unsigned long long t_begin = wall_clock64();
    unsigned long long t_closest = 0, t_shade = 0, t_shadow = 0;
    for (uint32_t bounce = 0; bounce <= fa.max_bounces; ++bounce)
    {
        const unsigned long long t0 = wall_clock64();
        if (lane == 0) row[bounce] = n;
        // ---- closest hits ----
        if (lane == 0) row[130] = 0;
        w4_trace_body<false, 12, false, true, true>(sc, fa.o4[cur], fa.d4[cur], (const uint32_t*)nullptr, n, (uint32_t*)nullptr, fa.hits, log, fa.spill,
            fa.tune, my_slow, row + 130, row + 128, (unsigned long long*)nullptr, 0u, 0u, fa.tail_q, 1u, stack, w, G, first, S);
        {
            __threadfence_block();
            const uint32_t n_slow = row[130];
            for (uint32_t k = lane; k < n_slow; k += 64u)                  // rays with a non-finite 1 / dir: the reference's loop on the exact BVH2
            {
                const uint32_t i = my_slow[k];
                float4 hit;
                (void)v1_trace_ray<false>(sc, fa.o4[cur][i], fa.d4[cur][i], spill_push, spill_pop, hit);
                fa.hits[i] = hit;
            }
            if (lane == 0 && n_slow != 0u) row[129] += n_slow;
            __threadfence_block();
        }
        const unsigned long long t1 = wall_clock64();
        t_closest += t1 - t0;
        // ---- Miss / HitSurface per entry; outgoing and shadow rays compacted into the wave's chunks of the next queues ----
        ShadeArgs a = fa.shade;
        a.in_o4 = fa.o4[cur]; a.in_d4 = fa.d4[cur]; a.in_thr = fa.thr[cur]; a.hits = fa.hits;
        a.bounce = bounce;
        a.emit_outgoing = (fa.drop_last && bounce >= fa.max_bounces) ? 0u : 1u;
        a.final_bounce = bounce >= fa.max_bounces ? 1u : 0u;
        a.count_in_ray = 1u;
        uint32_t n_next = 0, n_shadow = 0;
        for (uint32_t base = 0; base < n; base += 64u)
        {
            const uint32_t p = base + lane;
            bool want_shadow = false, want_next = false, no_block = false;
            uint32_t next_flag = 0, sh_entry = 0;
            float4 sh_o = make_float4(0, 0, 0, 0), sh_d = sh_o, nx_o = sh_o, nx_d = sh_o, nx_t = sh_o;
            if (p < n)
                shade_entry<FURNACE, BLUE, false, false>(sc, tile, a, index_of(p), want_shadow, want_next, no_block, next_flag, sh_o, sh_d, nx_o, nx_d, nx_t, sh_entry);
            const unsigned long long ms = __ballot(want_shadow), mn = __ballot(want_next);
            const unsigned long long lt = (1ull << lane) - 1ull;
            if (want_shadow)
            {
                const uint32_t i = index_of(n_shadow + (uint32_t)__popcll(ms & lt));
                fa.sh_o4[i] = sh_o; fa.sh_d4[i] = sh_d; fa.sh_aux[i] = sh_entry;
            }
            if (want_next)
            {
                const uint32_t i = index_of(n_next + (uint32_t)__popcll(mn & lt));
                fa.o4[cur ^ 1u][i] = nx_o; fa.d4[cur ^ 1u][i] = nx_d; fa.thr[cur ^ 1u][i] = nx_t;
            }
            n_shadow += (uint32_t)__popcll(ms);
            n_next += (uint32_t)__popcll(mn);
        }
        __threadfence_block();
        const unsigned long long t2 = wall_clock64();
        t_shade += t2 - t1;
        if (lane == 0) row[64u + bounce] = n_shadow;
        // ---- shadow rays: an occluded one retracts its path's tentative direct sample (AccumulateDirectSamples fused) ----
        if (n_shadow != 0u)
        {
            if (lane == 0) row[130] = 0;
            w4_trace_body<true, 12, false, true, true>(sc, fa.sh_o4, fa.sh_d4, fa.sh_aux, n_shadow, (uint32_t*)nullptr, (float4*)nullptr, log, fa.spill,
                fa.tune, my_slow, row + 130, row + 128, (unsigned long long*)nullptr, 0u, 0u, fa.tail_q, 1u, stack, w, G, first, S);
            __threadfence_block();
            const uint32_t n_slow = row[130];
            for (uint32_t k = lane; k < n_slow; k += 64u)
            {
                const uint32_t i = my_slow[k];
                const float4 rd = fa.sh_d4[i];
                float4 hit;
                if (v1_trace_ray<true>(sc, fa.sh_o4[i], rd, spill_push, spill_pop, hit)) log_retract(log, fa.sh_aux[i], __float_as_uint(rd.w));
            }
            if (lane == 0 && n_slow != 0u) row[129] += n_slow;
            __threadfence_block();
        }
        t_shadow += wall_clock64() - t2;
        cur ^= 1u;
        n = n_next;
        if (n == 0u) break;
    }
    if (lane == 0)
    {
        row[131] = (uint32_t)t_closest; row[132] = (uint32_t)t_shade; row[133] = (uint32_t)t_shadow; row[134] = (uint32_t)(wall_clock64() - t_begin);
    }
    // ---- the radiance log of the wave's pixels, replayed in the order the reference adds (k_flush for one sample in flight) ----
    for (uint32_t p = lane; p < n_pixels; p += 64u)
    {
        const uint32_t id = index_of(p);
        const uint32_t c = log.cnt[id];
        if (c == 0u) continue;
        float4 r = fa.radiance[id];
        for (uint32_t k = 0; k < c; ++k)
        {
            const rt_rgb v = *reinterpret_cast<const rt_rgb*>(log.rlog + 3 * ((size_t)k * log.stride + id));
            r.x += v.x; r.y += v.y; r.z += v.z;
        }
        log.cnt[id] = 0;
        fa.radiance[id] = r;
    }
}

// The per-bounce ray counters of a k_frame launch: column sums of the waves' rows into DCounters (what k_raygen's and k_shade's atomics leave
// after a sample of the stage kernels: queue[b], shadow[b]; the spill / slow statistics add up).
__global__ __launch_bounds__(256) void k_frame_sum(const uint32_t* __restrict__ wave_counts, uint32_t n_waves, uint32_t bounces, DCounters* __restrict__ counters)
{
    __shared__ uint32_t part[256];
    const uint32_t col = blockIdx.x;            // 0 .. 63 closest, 64 .. 127 shadow, 128 spills, 129 slow rays
    uint32_t s = 0;
    for (uint32_t wv = threadIdx.x; wv < n_waves; wv += 256u) s += wave_counts[(size_t)wv * RT_FRAME_COUNT_STRIDE + col];
    part[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t step = 128u; step != 0u; step >>= 1)
    {
        if (threadIdx.x < step) part[threadIdx.x] += part[threadIdx.x + step];
        __syncthreads();
    }
    if (threadIdx.x == 0)
    {
        if (col < 64u) { if (col <= bounces) counters->queue[col] = part[0]; }
        else if (col < 128u) { if (col - 64u <= bounces) counters->shadow[col - 64u] = part[0]; }
        else if (col == 128u) counters->stack_spills += part[0];
        else counters->slow_rays += part[0];
    }
}
