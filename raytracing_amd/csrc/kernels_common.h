// kernels_common.h -- device-side records, counters and the ray producer's helpers shared by
// the wavefront path-tracing kernels (hand-written for gfx950).  The kernels themselves:
//   raygen_kernels.h  trace_kernels.h  shade_kernels.h  aov_kernels.h  relayout_kernels.h
//
// Reference kernels replaced (src/kernels/cl/):
//   raygeneration.cl:65-139                        -> k_raygen
//   trace_bvh.cl:99-211                            -> k_trace_w4<false> (k_trace2<false>, k_trace_v1<false>)
//   trace_bvh.cl (-D SHADOW_RAYS) + accumulate_direct_samples.cl:27-53
//                                                  -> k_trace_w4<true>  (k_trace2<true>,  k_trace_v1<true>)
//   miss.cl:41-77 + hit_surface.cl:30-186 + clear_counter.cl (x2)
//                                                  -> k_shade
//   resolve_radiance.cl:31-86                      -> k_resolve
//   reset_radiance.cl / increment_counter.cl       -> hipMemsetAsync / host scalar
//
// Device data layout (HBM), chosen for coalesced 16-byte accesses:
//   ray queues   SoA: o4[i] = (origin.xyz, t_max), d4[i] = (dir.xyz, path id bits), thr[i] = (throughput.xyz,
//                log entries so far).  1/dir and the direction signs are computed by the CONSUMER when a lane starts
//                the ray (three IEEE divides per ray in the traversal kernel's refill phase): rounds 1-2 kept a third
//                16-byte record per ray for them, written by the producer -- 56 bytes of per-path state and one access
//                per ray more than this.
//                path id = slot * n_pixels + pixel, where `slot` numbers the samples in flight
//                (RT_OPT_SAMPLES_IN_FLIGHT); a shadow ray also carries the radiance-log entry
//                of its deferred direct sample in sh_aux[i]
//   radiance log per path: cnt[id] + log[k][id] (3 floats, 12 bytes) = the path's radiance
//                contributions in the order the reference adds them (miss or emission,
//                then direct light, per bounce).  k_flush replays them pixel by pixel,
//                sample by sample, so the fp32 sum is associated exactly as in the
//                reference although several samples are traced concurrently.
//   BVH          wnodes: one 64-byte record per folded piece of the reference BVH2 (a node and up to two more interior
//                nodes below it: 4 slots = the piece's frontier, 8-bit boxes on an exactly representable grid, per-octant
//                visit order): the tree k_trace_w4 walks (build_wide_bvh, rt_hip.hip);
//                nodes: one 64-byte "child-pair" record per INTERIOR node of the reference BVH2: both children's exact
//                boxes + refs in one line (k_trace2, k_trace_v1).  Topology, near/far rule and cull decisions are
//                exactly the reference's (trace_kernels.h).
//   trace tris   64 B, line aligned: (p1, last-in-leaf flag), e1 = p2-p1, e2 = p3-p1, spare
//   shade tris   128 B, line aligned: p1..p3, n1..n3, uv1..uv3, material
#pragma once
#include "device_math.h"
#include "rt_types.h"

#define RT_LEAF_BIT 0x80000000u
#define RT_EMPTY_REF 0xFFFFFFFFu
#define RT_TRACE_STACK_LDS 24     // per-lane stack entries kept in LDS
#define RT_TRACE_STACK_MAX 64     // the reference's nodesToVisit[64] (trace_bvh.cl:142)
// A shadow ray carries its path id in direction.w and the radiance-log entry of its deferred direct sample in sh_aux.

// one entry of the radiance log: RGB, 12 bytes (global_load/store_dwordx3)
struct rt_rgb { float x, y, z; };
RT_DEV void log_store(float* __restrict__ rlog, size_t index, float x, float y, float z)
{
    rt_rgb v = {x, y, z};
    *reinterpret_cast<rt_rgb*>(rlog + 3 * index) = v;
}

// Non-temporal queue accesses (round 6).  A trace launch reads every ray of its queue and writes every hit exactly once; marked non-temporal (the `nt` bit of
// global_load / global_store) those streams do not push the trees' records out of an XCD's 4 MB of L2.  Measured as build variants on the same box, two runs
// each (profiles/r06/call02.log, call03.log): the 10 M-triangle config, whose closest-hit walk misses the L2 for 45 % of its record fetches, 3785 / 3788 ->
// 3864 / 3870 Mrays/s (closest-hit trace 10.31 -> 10.02 ms per sample) and again 3766 / 3780 -> 3833 / 3859; the 2.8 M-triangle headline 6940 / 6946 ->
// 6915 / 6928 on one box, 6800 / 6812 -> 6871 / 6888 on the next.  The same hint on k_shade's streams moved nothing on either (dropped); on the leaf loop's
// triangle records it LOSES 9 % / 22 % (6308 / 2994: those records are re-read by the neighbouring rays of the wave and the next waves -- dropped).
typedef float rt_v4f __attribute__((ext_vector_type(4)));
RT_DEV float4 q_load(const float4* p) { const rt_v4f t = __builtin_nontemporal_load((const rt_v4f*)p); return make_float4(t.x, t.y, t.z, t.w); }
RT_DEV void q_store(float4* p, const float4& v) { const rt_v4f t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, (rt_v4f*)p); }

// Where the entries of a path's radiance log live.  FULL layout (inline_entries = 2 (B + 1), the worst case a path can
// log): entry k of path id at row-major index k * stride + id.  COMPACT layout (round 3; rt_integrate with many samples in
// flight): only the first inline_entries (6) rows exist for every path -- a path of the benchmark scene logs 2.7 entries on
// average, 1 % of the paths more than 6 -- and a path that is about to need a 7th gets an OVERFLOW BLOCK for its entries
// inline_entries .. 2 (B + 1) - 1 from a bump-allocated pool of ovf_blocks blocks (k_shade, one bounce ahead, one atomic per
// 512-path block): entry k >= inline_entries of a path with block `slot` at inline_entries * stride + (k - inline_entries) *
// ovf_blocks + slot.  A pool that runs dry raises DCounters::log_ovf_flag; the host then discards the batch and repeats it in
// the full layout (rt_integrate), so the sum is exact either way.
struct DLog
{
    float* rlog;                  // 3 floats per entry
    uint32_t* cnt;                // entries of path id (k_flush replays that many)
    uint32_t* ovf_slot;           // compact: the overflow block of path id (RT_EMPTY_REF: none); nullptr in the full layout
    uint32_t stride;              // paths per row
    uint32_t inline_entries;
    uint32_t ovf_blocks;
};
// index of entry k of path id (in entries), or ~0 when the entry has no home (pool dry: the batch is being discarded)
RT_DEV size_t log_index(const DLog& L, uint32_t k, uint32_t id, uint32_t slot)
{
    if (k < L.inline_entries) return (size_t)k * L.stride + id;
    if (slot >= L.ovf_blocks) return ~(size_t)0;
    return (size_t)L.inline_entries * L.stride + (size_t)(k - L.inline_entries) * L.ovf_blocks + slot;
}
RT_DEV void log_put(const DLog& L, uint32_t k, uint32_t id, uint32_t slot, float x, float y, float z)
{
    const size_t i = log_index(L, k, id, slot);
    if (i != ~(size_t)0) log_store(L.rlog, i, x, y, z);
}

struct DScene
{
    const float4* nodes;          // 4 x float4 per interior node
    const float4* tris_rt;        // 4 x float4 per triangle
    const float4* tris_sh;        // 8 x float4 per triangle
    const rt_packed_material* materials;
    const rt_texture* textures;
    const uint32_t* texture_data;
    const float4* lights;         // 3 x float4 per light: origin, radiance, (type bits,0,0,0)
    const float4* env;
    const float* gamma_lut;       // pow(byte / 255, 2.2f) for the 256 texel values (k_fill_gamma_lut)
    int env_w, env_h;
    uint32_t light_count;
    uint32_t root_ref;            // RT_LEAF_BIT | first triangle, or interior node 0
    uint32_t entry_ref;           // "super-root" record: child 0 = (root box, root_ref), child 1 empty
    const float4* wnodes;         // 4-wide quantized nodes (k_trace_w4), 4 x float4 each; nullptr = not built
    uint32_t w_entry_ref;         // wide node 0, or RT_LEAF_BIT | first triangle when the root is a leaf
    const float4* wnodes_sh;      // the tree the SHADOW rays walk: the backend's own over the reference's leaves (own_bvh.h), or wnodes
    uint32_t w_sh_entry_ref;
    float root_min[3];
    float root_max[3];
    // opt-in extensions (rt_scene_desc): nullptr / 0 = the reference's behaviour
    const uint16_t* mat_tex16;    // 6 texture indices per material (0xFFFF = none) replacing the packed 8-bit ones
    const uint32_t* emissive;     // emissive triangle indices (Scene::GetEmissiveIndices)
    uint32_t emissive_count;
    uint32_t emissive_nee;        // RT_SCENE_EMISSIVE_NEE
};

struct DTile                      // which pixels of the full image this frame owns
{
    uint32_t width, height;       // full image
    uint32_t band_h, rank, nranks;
    uint32_t local_rows;
};

RT_DEV uint32_t tile_global_row(const DTile& t, uint32_t ly)
{
    uint32_t band = ly / t.band_h;
    return (band * t.nranks + t.rank) * t.band_h + (ly - band * t.band_h);
}

#define RT_LOG_SUBPOOLS 64u        // the overflow pool of the compact log is handed out from this many bump counters (k_shade)
struct DCounters                  // one per frame, device memory
{
    uint32_t queue[64];           // queue[b]  = rays in the incoming queue of bounce b
    uint32_t shadow[64];          // shadow[b] = shadow rays emitted at bounce b
    unsigned long long total_closest, total_shadow, samples;
    uint32_t last_queue[64], last_shadow[64];
    // work-distribution heads of the persistent trace kernels: one per XCD and per
    // kernel flavour (0 = closest, 1 / 2 = shadow rays of an even / odd bounce: the shadow trace of bounce b may
    // still be running while k_shade of bounce b + 1 prepares the next one), offsets inside the XCD's region
    uint32_t head[3][8];
    // rays k_trace_w4 hands to the BVH2 kernel (non-finite 1/dir): list length and that launch's work heads
    uint32_t slow_count[4];       // per flavour ([3] unused)
    uint32_t slow_head[3][8];
    uint32_t stack_spills;        // lane-steps with stack entries in the HBM spill area (k_trace2 / k_trace_w4), since the last reset
    uint32_t slow_rays;           // rays k_trace_w4 handed to k_trace2, since the last reset
    // launch timeline of k_trace_w4<closest> per bounce (rt_frame_debug_timeline), 100 MHz wall clock:
    // first wave started, first wave found the queue dry, last wave left
    unsigned long long tl_start[64], tl_dry[64], tl_end[64];
    // ... of its slowest ray: most traversal steps (wide nodes + triangles) any ray took, and the ticks from hand-out to
    // retirement (high bits) and steps (low 24 bits) of the ray that took longest ...
    unsigned long long tl_ray_steps[64], tl_ray_ticks[64];
    // ... and when its waves left, in 25 us bins after the first wave found the queue dry (all recorded launches together)
    unsigned long long tl_exit_hist[64];
    // compact radiance log (DLog): next free overflow block of the batch in flight, and "the pool ran dry"
    uint32_t log_ovf_next[RT_LOG_SUBPOOLS], log_ovf_flag;
};

// ray_inv_dir and ray_sign of TraceBvh (trace_bvh.cl:125-129), packed as (inv.xyz, sign bits)
#define RT_SIGN_SLOW 8u
RT_DEV float4 ray_inverse(f3 dir)
{
    f3 inv = F3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
    uint32_t sign_bits = (inv.x < 0.0f ? 1u : 0u) | (inv.y < 0.0f ? 2u : 0u) | (inv.z < 0.0f ? 4u : 0u);
    // a non-finite component (dir component 0, denormal or NaN) can make the slab test produce
    // 0 * inf = NaN: such rays keep the select-form min/max of the reference (box_test).  So do rays with a
    // component of 1/dir beyond 2^96: k_trace_w4's one-fma slab distances must not overflow (trace_kernels.h).
    const float lim = 0x1p96f;
    if (!(__builtin_fabsf(inv.x) < lim && __builtin_fabsf(inv.y) < lim && __builtin_fabsf(inv.z) < lim))
        sign_bits |= RT_SIGN_SLOW;
    return make_float4(inv.x, inv.y, inv.z, __uint_as_float(sign_bits));
}
