// trace_kernels.h -- BVH traversal (trace_bvh.cl, + accumulate_direct_samples.cl for shadow rays):
// the slab / triangle tests, k_trace_v1 (per-ray loop: tiny launches), k_trace2 (exact BVH2 walk in three wave-uniform
// loops: the fallback and the hand-over kernel) and k_trace_w4 (4-wide quantised tree: the production kernel).
// Round 1's flat state-machine kernel and the packet kernel lost on every measured launch and were removed in round 3
// (their measurements: profiles/r01_*, r02_packet_kernel_on_coherent_bounces.log).
#pragma once
#include "kernels_common.h"

// ---------------------------------------------------------------------------
// BVH traversal (trace_bvh.cl:28-211)
// ---------------------------------------------------------------------------
// Exactness argument (DESIGN.md "traversal equivalence"): the reference pops a
// node, box-tests it against the CURRENT t_max, and descends near-first.  Here
// both children are box-tested when their parent is visited; the near child
// is visited next with the same t_max the reference would use; the far child is
// pushed with its entry distance A = max(max3(min(t0,t1)), t_min) and re-tested
// at pop time by `t_max >= A`, which (for the box test's min/max select forms
// and a non-increasing t_max) is equivalent to re-running the full box test.
// Leaves are visited in the reference's order and triangles are tested in
// array order with the same accept rule, so the closest hit (including ties,
// "later triangle replaces", trace_bvh.cl:157-162) is identical.

RT_DEV bool box_test(float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz, f3 org, f3 inv,
    float t_min, float t_max, float& entry)
{
    // RayBounds, trace_bvh.cl:85-97
    float t0x = (bminx - org.x) * inv.x, t0y = (bminy - org.y) * inv.y, t0z = (bminz - org.z) * inv.z;
    float t1x = (bmaxx - org.x) * inv.x, t1y = (bmaxy - org.y) * inv.y, t1z = (bmaxz - org.z) * inv.z;
    float lox = cl_min(t0x, t1x), loy = cl_min(t0y, t1y), loz = cl_min(t0z, t1z);
    float hix = cl_max(t0x, t1x), hiy = cl_max(t0y, t1y), hiz = cl_max(t0z, t1z);
    float tmin = cl_max(cl_max(cl_max(lox, loy), loz), t_min);
    float tmax = cl_min(cl_min(cl_min(hix, hiy), hiz), t_max);
    entry = tmin;
    return tmax >= tmin;
}

// The same test on v_min_f32 / v_max_f32 (v_min3 / v_max3): 20 VALU per child pair instead
// of 48 compare+select.  minNum/maxNum differ from the select forms above only (a) in the
// sign of a zero result -- every value here feeds comparisons only -- and (b) when an
// operand is NaN, which needs 0 * inf, i.e. a non-finite 1/dir component: rays with one are
// flagged by the producer (RT_SIGN_SLOW) and take box_test.
// (v_min/v_max are issued directly: through fminf/fmaxf the compiler first quiets every
// operand with a v_max_f32 x, x, eight extra instructions per child pair that only matter for
// signalling NaNs, which cannot occur here.)
RT_DEV float hw_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
RT_DEV float hw_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
RT_DEV float hw_min3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
RT_DEV float hw_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// Child-pair record (64 bytes, one per interior node of the reference BVH2), laid out so that the
// six (bound - origin) * inv_dir products of a child come out of packed-fp32 instructions without
// register shuffles (x and y of a corner share a 64-bit register pair with the ray's (ox, oy)):
//   q0 = (c0.min.x, c0.min.y, c0.max.x, c0.max.y)     q1 = (c1.min.x, c1.min.y, c1.max.x, c1.max.y)
//   q2 = (c0.min.z, c0.max.z, c1.min.z, c1.max.z)     q3 = (ref of child 0, ref of child 1, split axis, -)
#define RT_NODE_C0(q0, q1, q2) (q0).x, (q0).y, (q2).x, (q0).z, (q0).w, (q2).y
#define RT_NODE_C1(q0, q1, q2) (q1).x, (q1).y, (q2).z, (q1).z, (q1).w, (q2).w

RT_DEV bool box_test_fast(float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz, f3 org, f3 inv,
    float t_min, float t_max, float& entry)
{
    float t0x = (bminx - org.x) * inv.x, t0y = (bminy - org.y) * inv.y, t0z = (bminz - org.z) * inv.z;
    float t1x = (bmaxx - org.x) * inv.x, t1y = (bmaxy - org.y) * inv.y, t1z = (bmaxz - org.z) * inv.z;
    float lox = hw_min(t0x, t1x), loy = hw_min(t0y, t1y), loz = hw_min(t0z, t1z);
    float hix = hw_max(t0x, t1x), hiy = hw_max(t0y, t1y), hiz = hw_max(t0z, t1z);
    float tmin = hw_max(hw_max3(lox, loy, loz), t_min);
    float tmax = hw_min(hw_min3(hix, hiy, hiz), t_max);
    entry = tmin;
    return tmax >= tmin;
}

// AccumulateDirectSamples fused (accumulate_direct_samples.cl:46-52): k_shade logged the direct sample tentatively; an occluded
// shadow ray zeroes that entry (adding +0.0 is the identity).  Entries beyond the inline rows live in the path's overflow block.
RT_DEV void log_retract(const DLog& L, uint32_t entry, uint32_t id)
{
    const uint32_t slot = entry < L.inline_entries ? 0u : L.ovf_slot[id];
    log_put(L, entry, id, slot, 0.0f, 0.0f, 0.0f);
}

// One ray through the reference's loop (trace_bvh.cl:99-211) on the child-pair records: the per-ray walk of k_trace_v1, and the walk k_frame gives the
// rays its wide-tree body does not take (RT_SIGN_SLOW).  `push(sp, e)` / `pop(sp)` are the caller's stack (LDS + spill area, or the spill area alone).
// Returns occluded (SHADOW) / writes the closest hit.
template <bool SHADOW, class Push, class Pop>
RT_DEV bool v1_trace_ray(const DScene& sc, const float4 ro, const float4 rd, Push&& push, Pop&& pop, float4& hit_out)
{
    f3 org = F3(ro.x, ro.y, ro.z), dir = F3(rd.x, rd.y, rd.z);
    const float t_min = 0.0f;                                        // origin.w is 0 for every ray the path emits
    float t_max = ro.w;
    f3 inv = F3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);           // trace_bvh.cl:125
    uint32_t sign_bits = (inv.x < 0.0f ? 1u : 0u) | (inv.y < 0.0f ? 2u : 0u) | (inv.z < 0.0f ? 4u : 0u);

    uint32_t hit_prim = RT_INVALID_ID;
    float hit_u = 0.0f, hit_v = 0.0f, hit_t = 0.0f;
    bool occluded = false;

    int sp = 0;
    uint32_t ref = sc.root_ref;
    float entry;
    bool alive = box_test(sc.root_min[0], sc.root_min[1], sc.root_min[2], sc.root_max[0], sc.root_max[1],
        sc.root_max[2], org, inv, t_min, t_max, entry);

    while (alive)
    {
        bool need_pop;
        if (ref & RT_LEAF_BIT)
        {
            // leaf: test its triangles in array order (trace_bvh.cl:155-169)
            uint32_t prim = ref & ~RT_LEAF_BIT;
            bool last;
            do
            {
                const float4* tp = sc.tris_rt + (size_t)prim * 4;
                float4 a = tp[0], b = tp[1], cc = tp[2];
                last = a.w != 0.0f;
                f3 p1 = F3(a.x, a.y, a.z), e1 = F3(b.x, b.y, b.z), e2 = F3(cc.x, cc.y, cc.z);
                // RayTriangle, trace_bvh.cl:28-73
                f3 pvec = cross3(dir, e2);
                float det = dot3(e1, pvec);
                if (!(det < 1e-8f || -det > 1e-8f))
                {
                    float inv_det = 1.0f / det;
                    f3 tvec = org - p1;
                    float u = dot3(tvec, pvec) * inv_det;
                    if (!(u < 0.0f || u > 1.0f))
                    {
                        f3 qvec = cross3(tvec, e1);
                        float v = dot3(dir, qvec) * inv_det;
                        if (!(v < 0.0f || u + v > 1.0f))
                        {
                            float t = dot3(e2, qvec) * inv_det;
                            if (!(t < t_min || t > t_max))
                            {
                                hit_u = u; hit_v = v; hit_t = t; hit_prim = prim;
                                t_max = t;                           // :162
                                if (SHADOW) { occluded = true; }
                            }
                        }
                    }
                }
                ++prim;
            } while (!last && !(SHADOW && occluded));
            if (SHADOW && occluded) break;                           // goto endtrace, :164-167
            need_pop = true;
        }
        else
        {
            const float4* np = sc.nodes + (size_t)ref * 4;
            float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
            uint32_t c0 = __float_as_uint(n3.x), c1 = __float_as_uint(n3.y), axis = __float_as_uint(n3.z);
            float a0, a1;
            bool h0 = box_test(RT_NODE_C0(n0, n1, n2), org, inv, t_min, t_max, a0);
            bool h1 = box_test(RT_NODE_C1(n0, n1, n2), org, inv, t_min, t_max, a1);
            h1 = h1 && (c1 != RT_EMPTY_REF);
            // near child: first child unless the ray is negative along the split axis (:181-190)
            bool swap = (sign_bits >> axis) & 1u;
            uint32_t near_ref = swap ? c1 : c0, far_ref = swap ? c0 : c1;
            bool near_hit = swap ? h1 : h0, far_hit = swap ? h0 : h1;
            float far_entry = swap ? a0 : a1;
            if (near_hit)
            {
                if (far_hit)
                {
                    push(sp, make_uint2(far_ref, __float_as_uint(far_entry)));
                    ++sp;
                }
                ref = near_ref;
                need_pop = false;
            }
            else if (far_hit)
            {
                ref = far_ref;
                need_pop = false;
            }
            else
            {
                need_pop = true;
            }
        }
        if (need_pop)
        {
            alive = false;
            while (sp > 0)
            {
                --sp;
                uint2 e = pop(sp);
                if (t_max >= __uint_as_float(e.y))                   // box re-test at pop time
                {
                    ref = e.x;
                    alive = true;
                    break;
                }
            }
        }
    }
    hit_out = make_float4(hit_u, hit_v, __uint_as_float(hit_prim), hit_t);
    return occluded;
}

template <bool SHADOW>
__global__ __launch_bounds__(64) void k_trace_v1(DScene sc, const float4* __restrict__ o4, const float4* __restrict__ d4,
    const uint32_t* __restrict__ aux /* shadow rays: log entry of the deferred direct sample */, const uint32_t* __restrict__ count_ptr,
    float4* __restrict__ hits, DLog log, uint32_t /*force_sign_bits: v1 always uses box_test*/,
    uint2* __restrict__ spill)
{
    __shared__ uint2 stack[RT_TRACE_STACK_LDS][64];
    const uint32_t lane = threadIdx.x;
    const uint32_t count = *count_ptr;
    const uint32_t nchunks = (count + 63u) >> 6;
    // XCD-aware persistent schedule: block b runs on XCD b % 8 (observed
    // dispatch order); give each XCD one contiguous eighth of the queue so
    // that its private L2 sees one screen/queue region.
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const uint32_t cpx = (nchunks + 7u) >> 3;
    uint2* my_spill = spill + (size_t)(blockIdx.x * 64u + lane) * (RT_TRACE_STACK_MAX - RT_TRACE_STACK_LDS);
    auto push = [&](int sp, uint2 e) { if (sp < RT_TRACE_STACK_LDS) stack[sp][lane] = e; else my_spill[sp - RT_TRACE_STACK_LDS] = e; };
    auto pop = [&](int sp) -> uint2 { return (sp < RT_TRACE_STACK_LDS) ? stack[sp][lane] : my_spill[sp - RT_TRACE_STACK_LDS]; };

    for (uint32_t c = slot; c < cpx; c += per_xcd)
    {
        uint32_t chunk = xcd * cpx + c;
        uint32_t i = chunk * 64u + lane;
        if (i >= count) continue;

        float4 ro = o4[i], rd = d4[i];
        float4 hit;
        const bool occluded = v1_trace_ray<SHADOW>(sc, ro, rd, push, pop, hit);
        if (SHADOW)
        {
            // AccumulateDirectSamples (accumulate_direct_samples.cl:46-52) fused: k_shade
            // logged the direct sample tentatively; an occluded ray retracts it
            if (occluded)
            {
                log_retract(log, aux[i], __float_as_uint(rd.w));
            }
        }
        else
        {
            hits[i] = hit;
        }
    }
}


#define RT_TRACE_BATCH 128u        // ray indices a wave takes from its queue region per hand-out (k_trace2; k_trace_w4 picks its own)

// ---------------------------------------------------------------------------
// k_trace2: persistent traversal with SEPARATE wave-uniform loops (round 2)
// ---------------------------------------------------------------------------
// rocprofv3 on round 1's flat state-machine kernel (profiles/r01_final_pmc_summary.txt) shows one VALU issued per SIMD
// every 4.03 cycles -- the vector ALU's issue rate (tools/issue_microbench.hip) -- at ~124
// wave-instructions per ray, of which the box tests themselves need ~35.  The rest is the price
// of the single flat loop: every iteration walks through the ray-start, triangle and node code
// and their merges (exec-mask bookkeeping, phi moves) whenever ONE lane needs them.
// Here the three activities are three loops with wave-uniform (scalar) trip conditions:
//
//   A  retire finished rays + start new ones      (only when the node loop has run dry)
//   B  one triangle per waiting lane              (while >= leaf_q lanes wait at a leaf)
//   C  one child-pair node per lane               (while >= node_q lanes have a node)
//
// so the hot loop C contains nothing but the node fetch, the two slab tests, the push and
// the pop.  Lanes that reach a leaf or finish their ray sit out C until fewer than node_q
// lanes are left in it; then B and A serve everybody who waits, together.  The per-lane
// sequence of node visits, triangle tests and t_max updates is exactly k_trace_v1's (and the
// reference's): only the interleaving between lanes changes.
#define RT_IDLE_REF 0xFFFFFFFFu
typedef float rt_v2f __attribute__((ext_vector_type(2)));

// Both slab tests of a child-pair record on packed fp32: (bound - origin) * inv_dir for the 24
// planes in 12 v_pk_add_f32 / v_pk_mul_f32 (each an IEEE subtract / multiply per component, the
// reference's two roundings), then the v_min / v_max reduction of box_test_fast.
RT_DEV void pair_test_fast(const float4 q0, const float4 q1, const float4 q2, rt_v2f oxy, float oz, rt_v2f ixy, float iz,
    float t_min, float t_max, bool& h0, bool& h1, float& a0, float& a1)
{
    const rt_v2f ozz = {oz, oz}, izz = {iz, iz};          // one register each: the packed ops broadcast the low half
    rt_v2f c0a = ((rt_v2f){q0.x, q0.y} - oxy) * ixy;      // child 0: t0.x, t0.y
    rt_v2f c0b = ((rt_v2f){q0.z, q0.w} - oxy) * ixy;      //          t1.x, t1.y
    rt_v2f c1a = ((rt_v2f){q1.x, q1.y} - oxy) * ixy;      // child 1
    rt_v2f c1b = ((rt_v2f){q1.z, q1.w} - oxy) * ixy;
    rt_v2f c0z = ((rt_v2f){q2.x, q2.y} - ozz) * izz;      // child 0: t0.z, t1.z
    rt_v2f c1z = ((rt_v2f){q2.z, q2.w} - ozz) * izz;      // child 1
    float lo0 = hw_max3(hw_min(c0a.x, c0b.x), hw_min(c0a.y, c0b.y), hw_min(c0z.x, c0z.y));
    float hi0 = hw_min3(hw_max(c0a.x, c0b.x), hw_max(c0a.y, c0b.y), hw_max(c0z.x, c0z.y));
    float lo1 = hw_max3(hw_min(c1a.x, c1b.x), hw_min(c1a.y, c1b.y), hw_min(c1z.x, c1z.y));
    float hi1 = hw_min3(hw_max(c1a.x, c1b.x), hw_max(c1a.y, c1b.y), hw_max(c1z.x, c1z.y));
    a0 = hw_max(lo0, t_min);
    a1 = hw_max(lo1, t_min);
    h0 = hw_min(hi0, t_max) >= a0;
    h1 = hw_min(hi1, t_max) >= a1;
}

// Wave-level pool of ray indices for the persistent kernels: RT_TRACE_BATCH indices at a time from the
// head of this XCD's region of the queue (one atomic), stealing from the next regions when it runs dry.
struct RayPool { uint32_t next, end, regions_tried; bool exhausted; };     // wave-uniform

// gives every lane with `wants` (and no ray yet) the next index, until all are served or the queue is dry
RT_DEV void hand_out_rays(RayPool& p, uint32_t lane, uint32_t xcd, uint32_t per, uint32_t count, uint32_t* __restrict__ heads,
    bool wants, uint32_t& ray_i, uint32_t batch = RT_TRACE_BATCH)
{
    unsigned long long need = __ballot(wants && ray_i == RT_INVALID_ID);
    while (need)
    {
        if (p.next >= p.end)
        {
            bool got = false;
            while (p.regions_tried < 8u)
            {
                uint32_t x = (xcd + p.regions_tried) & 7u;
                uint32_t rb = x * per < count ? x * per : count;
                uint32_t re = (x + 1u) * per < count ? (x + 1u) * per : count;
                uint32_t b = 0;
                if (lane == 0 && rb < re) b = atomicAdd(&heads[x], batch);
                b = __shfl(b, 0, 64);
                if (rb < re && b < re - rb)
                {
                    p.next = rb + b;
                    p.end = (b + batch < re - rb) ? rb + b + batch : re;
                    got = true;
                    break;
                }
                ++p.regions_tried;
            }
            if (!got) { p.exhausted = true; break; }
        }
        uint32_t avail = p.end - p.next;
        uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
        uint32_t n = (uint32_t)__popcll(need);
        if (wants && ray_i == RT_INVALID_ID && rank < avail) ray_i = p.next + rank;
        p.next += n < avail ? n : avail;
        need = __ballot(wants && ray_i == RT_INVALID_ID);
    }
}

template <bool SHADOW, int STACK>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(STACK <= 10 ? 8 : 6, 8))) void k_trace2(DScene sc, const float4* __restrict__ o4, const float4* __restrict__ d4,
    const uint32_t* __restrict__ aux /* shadow rays: log entry of the deferred direct sample */, const uint32_t* __restrict__ count_ptr,
    uint32_t* __restrict__ heads,
    float4* __restrict__ hits,
    DLog log, uint32_t force_sign_bits, uint2* __restrict__ spill, uint32_t tune,
    const uint32_t* __restrict__ index_list /* nullptr: the whole queue; else queue entries to trace, *count_ptr of them */,
    uint32_t* __restrict__ spill_count /* statistics: pushes beyond the LDS stack */)
{
    __shared__ uint2 stack[STACK][64];
    uint32_t* const stack32 = reinterpret_cast<uint32_t*>(&stack[0][0]);     // SHADOW: 2 * STACK entries of 4 bytes
    // spill area: plain stores, but VOLATILE loads -- otherwise the compiler folds the LDS and the HBM side of a pop into
    // one flat_load on a selected generic address (and the common LDS pop loses its ds_read_b64)
    uint2* const vspill = spill;
    uint32_t* const vspill32 = reinterpret_cast<uint32_t*>(spill);
    const volatile uint2* const lspill = spill;
    const volatile uint32_t* const lspill32 = reinterpret_cast<const volatile uint32_t*>(spill);
    const uint32_t lane = threadIdx.x;
    const uint32_t count = *count_ptr;
    if (count == 0) return;
    const uint32_t node_q = tune & 0xFFu, leaf_q = (tune >> 8) & 0xFFu;
    const uint32_t xcd = blockIdx.x & 7u;
    const uint32_t per = (((count + 7u) >> 3) + 63u) & ~63u;                 // one 64-aligned eighth of the queue per XCD
    const uint32_t spill_base = (blockIdx.x * 64u + lane) * (uint32_t)(RT_TRACE_STACK_MAX - STACK);
    const char* const node_base = reinterpret_cast<const char*>(sc.nodes);   // 32-bit byte offsets: the arrays
    const char* const tri_base = reinterpret_cast<const char*>(sc.tris_rt);  // stay below 4 GiB (rt_scene_upload)

    RayPool pool = {0u, 0u, 0u, false};
    uint32_t ref = RT_IDLE_REF;                                              // interior node | RT_LEAF_BIT + triangle | idle
    uint32_t ray_i = RT_INVALID_ID;                                          // != invalid while a result is owed
    uint32_t sign_bits = 0, hit_prim = RT_INVALID_ID;
    int sp = 0;
    uint32_t n_spills = 0;                                                   // statistics (wave-uniform): lane-steps with entries in the HBM spill area
    rt_v2f oxy = {0.0f, 0.0f}, ixy = oxy;                                    // origin.xy and (1/dir).xy as register pairs
    float oz = 0.0f, iz = 0.0f;
    f3 dir = F3s(0.0f);
    float t_max = 0.0f, hit_u = 0.0f, hit_v = 0.0f;
    uint32_t payload = 0, log_entry = 0;
    const float t_min = 0.0f;

    // next entry of this lane's stack that still passes the box test, or idle (ray finished)
    auto pop = [&]()
    {
        ref = RT_IDLE_REF;
        if (SHADOW)
        {
            if (sp > 0)
            {
                --sp;
                if (sp < 2 * STACK) ref = stack32[sp * 64 + lane];
                else ref = lspill32[(size_t)2 * spill_base + (uint32_t)(sp - 2 * STACK)];
            }
        }
        else
            while (sp > 0)
            {
                --sp;
                uint32_t ex, ey;
                if (sp < STACK) { const uint2 e = stack[sp][lane]; ex = e.x; ey = e.y; }
                else { ex = lspill[(size_t)spill_base + (uint32_t)(sp - STACK)].x; ey = lspill[(size_t)spill_base + (uint32_t)(sp - STACK)].y; }
                if (t_max >= __uint_as_float(ey)) { ref = ex; break; }       // box re-test at pop time
            }
    };

    for (;;)
    {
        // ---- A: retire finished rays, start new ones -------------------------------------
        if (__ballot(ref == RT_IDLE_REF) != 0ull)
        {
            if (ref == RT_IDLE_REF && ray_i != RT_INVALID_ID)
            {
                if (SHADOW)
                {
                    // AccumulateDirectSamples fused: an occluded ray retracts its tentative direct sample
                    if (hit_prim != RT_INVALID_ID)
                        log_retract(log, log_entry, payload);
                }
                else
                    hits[ray_i] = make_float4(hit_u, hit_v, __uint_as_float(hit_prim), t_max);
                ray_i = RT_INVALID_ID;
            }
            if (!pool.exhausted)
            {
                hand_out_rays(pool, lane, xcd, per, count, heads, ref == RT_IDLE_REF, ray_i);
                if (ref == RT_IDLE_REF && ray_i != RT_INVALID_ID)
                {
                    // ray start: 1/dir and the sign bits come from the producer; the root box test is
                    // the ordinary node test of the super-root record (rt_scene_upload)
                    if (index_list) ray_i = index_list[ray_i];
                    const float4 q0 = o4[ray_i], q1 = d4[ray_i];
                    if (SHADOW) { payload = __float_as_uint(q1.w); log_entry = aux[ray_i]; }
                    oxy = (rt_v2f){q0.x, q0.y}; oz = q0.z;
                    dir = F3(q1.x, q1.y, q1.z);
                    t_max = q0.w;
                    const float4 q2 = ray_inverse(dir);                      // trace_bvh.cl:125-129
                    ixy = (rt_v2f){q2.x, q2.y}; iz = q2.z;
                    sign_bits = (__float_as_uint(q2.w) & 0xFFu) | force_sign_bits;
                    hit_prim = RT_INVALID_ID;
                    hit_u = 0.0f; hit_v = 0.0f;
                    sp = 0;
                    ref = sc.entry_ref;
                }
            }
        }
        if (__ballot(ref != RT_IDLE_REF) == 0ull) break;                    // queue dry and every ray retired
        // ---- B: triangles (trace_bvh.cl:28-73,155-169), one per waiting lane per pass ----
        {
            unsigned long long leaf_m = __ballot((int)ref < -1);
            const uint32_t n_node = (uint32_t)__popcll(__ballot((int)ref >= 0));
            if (leaf_m != 0ull && ((uint32_t)__popcll(leaf_m) >= leaf_q || n_node < node_q))
                do
                {
                    if ((int)ref < -1)
                    {
                        const uint32_t prim = ref & ~RT_LEAF_BIT;
                        const float4* tp = reinterpret_cast<const float4*>(tri_base + (size_t)(prim << 6));
                        const float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
                        const bool last = q0.w != 0.0f;
                        bool accepted = false;
                        const f3 org = F3(oxy.x, oxy.y, oz);
                        f3 p1 = F3(q0.x, q0.y, q0.z), e1 = F3(q1.x, q1.y, q1.z), e2 = F3(q2.x, q2.y, q2.z);
                        f3 pvec = cross3(dir, e2);
                        float det = dot3(e1, pvec);
                        if (!(det < 1e-8f || -det > 1e-8f))
                        {
                            float inv_det = 1.0f / det;
                            f3 tvec = org - p1;
                            float u = dot3(tvec, pvec) * inv_det;
                            if (!(u < 0.0f || u > 1.0f))
                            {
                                f3 qvec = cross3(tvec, e1);
                                float v = dot3(dir, qvec) * inv_det;
                                if (!(v < 0.0f || u + v > 1.0f))
                                {
                                    float t = dot3(e2, qvec) * inv_det;
                                    if (!(t < t_min || t > t_max))
                                    {
                                        hit_u = u; hit_v = v; hit_prim = prim;
                                        t_max = t;                           // :162
                                        accepted = true;
                                    }
                                }
                            }
                        }
                        if (SHADOW && accepted) ref = RT_IDLE_REF;           // goto endtrace, :164-167
                        else if (last) pop();
                        else ref = ref + 1u;
                    }
                    leaf_m = __ballot((int)ref < -1);
                } while ((uint32_t)__popcll(leaf_m) >= leaf_q && leaf_m != 0ull);
        }

        // ---- C: interior nodes, the hot loop ----------------------------------------------
        for (;;)
        {
            const unsigned long long node_m = __ballot((int)ref >= 0);
            if (node_m == 0ull) break;
            if ((uint32_t)__popcll(node_m) < node_q)
            {
                // leave when somebody who waits can be served: a lane at a leaf, or a finished lane
                // while the queue still has rays
                const unsigned long long waiting = __ballot((int)ref < -1 || (ref == RT_IDLE_REF && !pool.exhausted));
                if (waiting != 0ull) break;
            }
            if ((int)ref >= 0)
            {
                const float4* np = reinterpret_cast<const float4*>(node_base + (size_t)(ref << 6));
                const float4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
                const uint32_t c0 = __float_as_uint(q3.x), c1 = __float_as_uint(q3.y), axis = __float_as_uint(q3.z);
                float a0, a1;
                bool h0, h1;
                if (sign_bits & RT_SIGN_SLOW)
                {
                    const f3 org = F3(oxy.x, oxy.y, oz), inv = F3(ixy.x, ixy.y, iz);
                    h0 = box_test(RT_NODE_C0(q0, q1, q2), org, inv, t_min, t_max, a0);
                    h1 = box_test(RT_NODE_C1(q0, q1, q2), org, inv, t_min, t_max, a1);
                }
                else
                    pair_test_fast(q0, q1, q2, oxy, oz, ixy, iz, t_min, t_max, h0, h1, a0, a1);
                // near child: first child unless the ray is negative along the split axis (:181-190).
                // The hit flags are lane masks: they are swapped with scalar mask arithmetic and come
                // back as branch conditions (inverse ballot), no vector instruction involved.
                const bool swap = ((sign_bits >> axis) & 1u) != 0u;
                const unsigned long long m_h0 = __ballot(h0), m_h1 = __ballot(h1 && c1 != RT_EMPTY_REF), m_sw = __ballot(swap);
                const bool near_hit = __builtin_amdgcn_inverse_ballot_w64((m_sw & m_h1) | (~m_sw & m_h0));
                const bool far_hit = __builtin_amdgcn_inverse_ballot_w64((m_sw & m_h0) | (~m_sw & m_h1));
                const uint32_t near_ref = swap ? c1 : c0, far_ref = swap ? c0 : c1;
                if (near_hit & far_hit)
                {
                    if (SHADOW)
                    {
                        if (sp < 2 * STACK) stack32[sp * 64 + lane] = far_ref;
                        else vspill32[(size_t)2 * spill_base + (uint32_t)(sp - 2 * STACK)] = far_ref;
                    }
                    else
                    {
                        const uint32_t far_entry = __float_as_uint(swap ? a0 : a1);
                        if (sp < STACK) stack[sp][lane] = make_uint2(far_ref, far_entry);
                        else
                        {
                            vspill[(size_t)spill_base + (uint32_t)(sp - STACK)] = make_uint2(far_ref, far_entry);
                        }
                    }
                    ++sp;
                }
                if (near_hit) ref = near_ref;
                else if (far_hit) ref = far_ref;
                else pop();
            }
            n_spills += (uint32_t)__popcll(__ballot(sp > (SHADOW ? 2 * STACK : STACK)));
        }
    }
    if (lane == 0 && n_spills != 0u) atomicAdd(spill_count, n_spills);
}

// ---------------------------------------------------------------------------
// k_trace_w4: traversal of the 4-wide quantized tree (build_wide_bvh, rt_hip.hip)
// ---------------------------------------------------------------------------
// k_trace2 moves ~137 sixteen-byte L1 accesses per ray and the CU's L1 retires ~1.05 of them per
// clock whatever the instruction count (profiles/r02_ktrace2_pmc.txt: VALU -43 %, time -7 %): the
// kernel is bound by the NUMBER of accesses and of dependent round trips.  The wide tree halves
// both: one 64-byte record holds four descendants of a BVH2 node (the frontier of up to three folded interior nodes,
// picked by SAH: build_wide_bvh) as 8-bit boxes, so a ray makes fewer than half as many node visits and each visit costs
// the same four accesses.
//
// Exactness (see build_wide_bvh): interior boxes contain the reference's boxes, dequantise EXACTLY, and
// their slab distances are evaluated with one fma per plane plus an outward margin that covers the
// difference to the reference's own expression (loop C below), so culling is conservative;
// every leaf is re-tested with its exact fp32 bounds and the current t_max when it is reached, in the
// reference's depth-first near/far order; shadow rays only need the boolean, which no superset can
// change.  Rays whose 1/dir has a non-finite or huge component (RT_SIGN_SLOW: NaNs break the
// monotonicity argument, huge values could overflow) are not traced here: their queue indices go to
// `slow_list` and k_trace2 runs over that list afterwards.
//   stack entry: (ref, entry distance) as in k_trace2; up to three pushes per visit.
#define RT_LEAF_CONT_BIT 0x40000000u      // the leaf's box has been tested: this is its 2nd+ triangle
#define RT_W4_STACK_MAX 104               // <= 3 pending slots per wide level, <= 33 wide levels + slack

// Reads from the per-lane HBM spill area of the traversal stack (rare path).  Inline assembly on purpose: see the note
// at the use in k_trace_w4.  The wait inside covers every outstanding vector-memory operation of the wave.
RT_DEV uint2 spill_load64(const uint2* p)
{
    unsigned long long v;
    asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}
RT_DEV uint32_t spill_load32(const uint32_t* p)
{
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// One visit of a wide node (the body of k_trace_w4's loop C; also timed on its own by tools/visit_microbench.hip):
// the four slots' conservative slab tests and, for closest-hit rays, the reference's visit order.  Out: r[k] / e[k] =
// ref and entry distance of the slot visited k-th, e[k] = +inf for a slot that is empty or missed.
template <bool SHADOW>
RT_DEV void w4_test_slots(const float4 q0, const float4 q1, const float4 q2, const float4 q3, const f3 org, const f3 inv,
    const uint32_t sign_bits, const uint32_t octant4, const float t_min, const float t_max, uint32_t (&r)[4], float (&e)[4])
{
    const float INF = __builtin_inff();
    const uint32_t meta = __float_as_uint(q0.w);
    const float cx = __uint_as_float((meta & 0xFFu) << 23), cy = __uint_as_float(((meta >> 8) & 0xFFu) << 23),
                cz = __uint_as_float(((meta >> 16) & 0xFFu) << 23);
    // near / far plane words per axis, chosen by the ray's direction sign
    const bool nx = (sign_bits & 1u) != 0u, ny = (sign_bits & 2u) != 0u, nz = (sign_bits & 4u) != 0u;
    const uint32_t lox = __float_as_uint(q1.x), loy = __float_as_uint(q1.y), loz = __float_as_uint(q1.z);
    const uint32_t hix = __float_as_uint(q1.w), hiy = __float_as_uint(q2.x), hiz = __float_as_uint(q2.y);
    const uint32_t nwx = nx ? hix : lox, fwx = nx ? lox : hix;
    const uint32_t nwy = ny ? hiy : loy, fwy = ny ? loy : hiy;
    const uint32_t nwz = nz ? hiz : loz, fwz = nz ? loz : hiz;
    // Slab distance of grid plane q along an axis: the reference's expression on the dequantised plane is
    // E(q) = fl(fl(fl(q * cell + origin) - org) * inv) (the inner fl is exact, build_wide_bvh), monotone in q, so
    // "leaf box passes => stored box passes" holds for E exactly.  Evaluated here as one fma per plane,
    // F(q) = fl(q * a + b) with a = cell * inv (exact: cell is a power of two), b = fl(fl(origin - org) * inv):
    // both round the same real number T(q) = (q * cell + origin - org) * inv, |E - T| <= 2.1 u M and
    // |F - T| <= 4 u M with u = 2^-24 and M = 255 |a| + |b| >= |q a| + |b| -- so the near planes take
    // b - 2^-20 M and the far planes b + 2^-20 M (16 u M, plus 2^-120 against results in the denormal range):
    // F_near <= E_near and F_far >= E_far, the stored box only ever grows, interior culling only ever visits
    // MORE.  Leaves are re-tested with the reference's expression when they are reached, as before.
    // No overflow: |inv| < 2^96 (ray_inverse: other rays are RT_SIGN_SLOW), cell <= 2^20 and |origin| < 2^28
    // (build_wide_bvh), |org| < 2^29 (checked when the ray starts) => |q a| + |b| < 2^127.
    const float ax = cx * inv.x, ay = cy * inv.y, az = cz * inv.z;
    const float bx = (q0.x - org.x) * inv.x, by = (q0.y - org.y) * inv.y, bz = (q0.z - org.z) * inv.z;
    const float mx = __builtin_fmaf(255.0f, __builtin_fabsf(ax), __builtin_fabsf(bx)) + 0x1p-100f;
    const float my = __builtin_fmaf(255.0f, __builtin_fabsf(ay), __builtin_fabsf(by)) + 0x1p-100f;
    const float mz = __builtin_fmaf(255.0f, __builtin_fabsf(az), __builtin_fabsf(bz)) + 0x1p-100f;
    const float bnx = __builtin_fmaf(-0x1p-20f, mx, bx), bfx = __builtin_fmaf(0x1p-20f, mx, bx);
    const float bny = __builtin_fmaf(-0x1p-20f, my, by), bfy = __builtin_fmaf(0x1p-20f, my, by);
    const float bnz = __builtin_fmaf(-0x1p-20f, mz, bz), bfz = __builtin_fmaf(0x1p-20f, mz, bz);
    r[0] = __float_as_uint(q2.z); r[1] = __float_as_uint(q2.w); r[2] = __float_as_uint(q3.x); r[3] = __float_as_uint(q3.y);
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
        const float tnx = __builtin_fmaf((float)((nwx >> (8 * k)) & 0xFFu), ax, bnx);
        const float tny = __builtin_fmaf((float)((nwy >> (8 * k)) & 0xFFu), ay, bny);
        const float tnz = __builtin_fmaf((float)((nwz >> (8 * k)) & 0xFFu), az, bnz);
        const float tfx = __builtin_fmaf((float)((fwx >> (8 * k)) & 0xFFu), ax, bfx);
        const float tfy = __builtin_fmaf((float)((fwy >> (8 * k)) & 0xFFu), ay, bfy);
        const float tfz = __builtin_fmaf((float)((fwz >> (8 * k)) & 0xFFu), az, bfz);
        const float entry = hw_max(hw_max3(tnx, tny, tnz), t_min);
        const float exit = hw_min(hw_min3(tfx, tfy, tfz), t_max);
        e[k] = (exit >= entry && r[k] != RT_EMPTY_REF) ? entry : INF;      // INF = slot not visited
    }
    if (!SHADOW)
    {
        // the reference's order: depth-first over the BVH2 nodes this record folds, near child first at each of them
        // (trace_bvh.cl:181-190).  The folded subtree has any of the five shapes a binary tree with four leaves can have
        // (build_wide_bvh picks the frontier by SAH); the record stores the slots where these four conditional exchanges
        // can produce every order its shape asks for, and their settings per direction octant (4 bits each).
        const uint32_t sw = __float_as_uint(q3.z) >> octant4;
        uint32_t tr; float te; bool p;
        p = (sw & 1u) != 0u; tr = p ? r[1] : r[0]; r[1] = p ? r[0] : r[1]; r[0] = tr;  te = p ? e[1] : e[0]; e[1] = p ? e[0] : e[1]; e[0] = te;
        p = (sw & 2u) != 0u; tr = p ? r[3] : r[2]; r[3] = p ? r[2] : r[3]; r[2] = tr;  te = p ? e[3] : e[2]; e[3] = p ? e[2] : e[3]; e[2] = te;
        p = (sw & 4u) != 0u; tr = p ? r[2] : r[0]; r[2] = p ? r[0] : r[2]; r[0] = tr;  te = p ? e[2] : e[0]; e[2] = p ? e[0] : e[2]; e[0] = te;
        p = (sw & 8u) != 0u; tr = p ? r[3] : r[1]; r[3] = p ? r[1] : r[3]; r[1] = tr;  te = p ? e[3] : e[1]; e[3] = p ? e[1] : e[3]; e[1] = te;
    }
}

// Of the slots that pass their box test the FIRST in visit order is visited next and only the later ones go to the stack
// (round 2 pushed positions 3..1 and popped the first passing one right back: with 1.2 of 4 slots passing per visit that
// was 18.4 pushes per closest-hit ray instead of 8.6; same sequence of nodes -- an entry popped right after its push always
// passes the pre-cull: entry <= exit <= t_max -- shown on the CPU by the restatement in oracle/oracle.c,
// tests/test_wide_traversal_oracle.py; on the GPU: profiles/r03_call01_direct_variant_*).
// TAIL: the instance with loop D (below) for launches the host KNOWS to be small (at most RT_OPT_SMALL_LAUNCH_PATHS paths in the
// batch: every launch of the reference's one-sample-per-frame pattern).  It is an instance of its own because the extra loop
// costs the three hot loops their register allocation: 75 VGPRs (6 waves per SIMD instead of 7), or 72 with five dwords of
// phase A spilled -- either way ~2 % of a large launch (profiles/r04_call04_kernel_ab.log), which has no use for loop D.
// PRIVATE (round 5, k_frame below): the body as ONE WAVE's walk over a queue of its own -- rays 0 .. count - 1 of the wave live in ITS chunks of the
// arrays (position p at chunk priv_first + (p >> 6) * priv_stride, slot p & 63), handed out to idle lanes in order, no atomics, nothing shared; a
// ray the wide walk does not take (RT_SIGN_SLOW) comes back through `slow` for the caller to trace.  block_id / grid_blocks stand in for blockIdx.x /
// gridDim.x (the kernel below passes them; a wave of k_frame passes what gives it its own spill area).
template <bool SHADOW, int STACK, bool TIMELINE, bool TAIL, bool PRIVATE>
RT_DEV void w4_trace_body(const DScene& sc, const float4* __restrict__ o4, const float4* __restrict__ d4,
    const uint32_t* __restrict__ aux /* shadow rays: log entry of the deferred direct sample */, const uint32_t count,
    uint32_t* __restrict__ heads,
    float4* __restrict__ hits, const DLog& log, uint2* __restrict__ spill, uint32_t tune,
    uint32_t* __restrict__ slow_list, uint32_t* __restrict__ slow_count, uint32_t* __restrict__ stat_counts /* [0] spills, [1] slow rays */,
    unsigned long long* __restrict__ timeline /* TIMELINE: DCounters::tl_start + timeline_slot (rt_frame_debug_timeline) */,
    uint32_t timeline_slot, uint32_t chunk_below /* launches of fewer rays run in chunk mode (below) */,
    uint32_t tail_q /* loop D: with this many or fewer lanes busy and nothing to refill the others with, one fused pass serves all */,
    uint32_t chunk_refill /* chunk mode: idle lanes take the next rays of the wave's OWN chunks at once (below) */,
    uint2 (*stack)[64] /* LDS, STACK entries per lane */, const uint32_t block_id, const uint32_t grid_blocks,
    const uint32_t priv_first, const uint32_t priv_stride)
{
    uint32_t* const stack32 = reinterpret_cast<uint32_t*>(&stack[0][0]);     // SHADOW: 2 * STACK entries of 4 bytes
    // spill area: plain (cached) stores; the loads go through spill_load*: written in C they are folded with the LDS side
    // of the pop into one flat_load on a selected generic address, and the common LDS pop loses its ds_read_b64
    uint2* const vspill = spill;
    uint32_t* const vspill32 = reinterpret_cast<uint32_t*>(spill);
    const uint32_t lane = threadIdx.x;
    if (count == 0) return;
    bool dry_noted = false;
    if (TIMELINE && lane == 0) atomicMin(&timeline[0], wall_clock64());
    uint32_t tl_steps = 0, tl_max_steps = 0;                                 // TIMELINE only
    unsigned long long tl_t0 = 0, tl_max_ticks = 0, tl_steps_of_slowest = 0;
    const uint32_t node_q = tune & 0xFFu, leaf_q = (tune >> 8) & 0xFFu;
    // Grid sized from the LIVE counter (the reference launches width x height work-items whatever the queue holds:
    // "@TODO: use indirect dispatch", cl_pt_integrator.cpp:534,575).  The host launches the residency-sized persistent grid
    // without knowing `count`; with few rays (one sample per pixel in flight, late bounces) every lane of that grid gets a
    // handful of rays and the whole launch is its tail (DESIGN.md "Where a launch's time goes"): tune bits 24..31 = the
    // fewest rays per lane a wave is worth starting for, waves beyond that leave at once and the rest see a grid of n_blocks.
    uint32_t n_blocks = grid_blocks;
    if (!PRIVATE && (tune >> 24) != 0u)
    {
        const uint32_t want = ((count / (64u * (tune >> 24)) + 7u) & ~7u);
        n_blocks = want < 8u ? 8u : (want < n_blocks ? want : n_blocks);
        if (block_id >= n_blocks) return;
    }
    const uint32_t xcd = block_id & 7u;
    const uint32_t per = (((count + 7u) >> 3) + 63u) & ~63u;
    // rays a wave takes from the queue per hand-out: large, because every hand-out is one atomic on one of eight
    // addresses that 6000 waves share (128 -> 512 rays: +3 %, profiles/r02_handout_sweep.log), but never so large that a
    // wave would empty its region in fewer than ~4 hand-outs (small launches: late bounces, chunks, tiles)
    // CHUNK mode (tune bit 23): a wave takes 64 rays, finishes ALL of them, then takes the next 64 -- no refill of single
    // lanes.  Lane utilisation inside a chunk falls as its rays finish, but the rays in flight when the queue runs dry are
    // ordinary rays, not the long ones a refilling wave accumulates (length-biased sampling: DESIGN.md "Where a launch's time
    // goes"), so a launch ends one chunk after its queue does: the mode for launches that are all tail (one sample per pixel in flight)
    // -- decided here, from the live queue counter (the host does not know it): small launches (a frame's single sample per
    // pixel, the late bounces of any batch) are all tail in refill mode; 2430 vs 849 Mrays/s at one 1080p sample in flight,
    // equal at ~8, 3340 vs 5940 at 64 (profiles/r03_call05_chunk_vs_refill_cfg4.log).
    const bool chunk_mode = PRIVATE || (tune & 0x800000u) != 0u || count < chunk_below;
    uint32_t grab = ((tune >> 16) & 0x7Fu) ? ((tune >> 16) & 0x7Fu) * 16u : 512u;
    if (chunk_mode) grab = 64u;
    {
        const uint32_t fair = (per / ((n_blocks >> 3) * 4u + 1u)) & ~63u;
        grab = fair < grab ? (fair < 64u ? 64u : fair) : grab;
    }
    // (Tapering the hand-outs towards the end of the region does not shorten the tail of a launch -- 0.75-0.95 ms
    // after the first wave finds the queue dry, tools/launch_timeline.py -- by more than 0.1 ms: the tail is single
    // long rays on nearly empty waves, not the size of the last hand-outs.  profiles/r02_taper_sweep.log)
    const uint32_t spill_base = (block_id * 64u + lane) * (uint32_t)(RT_W4_STACK_MAX - STACK);
    const char* const node_base = reinterpret_cast<const char*>(SHADOW ? sc.wnodes_sh : sc.wnodes);
    const char* const tri_base = reinterpret_cast<const char*>(sc.tris_rt);
    const uint32_t entry_ref = SHADOW ? sc.w_sh_entry_ref : sc.w_entry_ref;

    RayPool pool = {0u, 0u, 0u, false};
    uint32_t chunk_next = PRIVATE ? 0u : block_id >> 3;   // chunk mode: this wave's next chunk of its XCD's region (PRIVATE: its next position)
    uint32_t ref = RT_IDLE_REF;                    // wide node | RT_LEAF_BIT (| RT_LEAF_CONT_BIT) + triangle | idle
    uint32_t ray_i = RT_INVALID_ID;
    uint32_t sign_bits = 0, octant4 = 0, hit_prim = RT_INVALID_ID;          // octant4: shift of this ray's entry in a node's order table
    int sp = 0;
    uint32_t n_spills = 0;                                                   // statistics (wave-uniform): lane-steps with entries in the HBM spill area
    f3 org = F3s(0.0f), dir = F3s(0.0f), inv = F3s(0.0f);
    float t_max = 0.0f, hit_u = 0.0f, hit_v = 0.0f;
    uint32_t payload = 0, log_entry = 0;
    const float t_min = 0.0f;
    const float INF = __builtin_inff();

    auto push = [&](uint32_t r, float entry)
    {
        if (SHADOW)
        {
            if (sp < 2 * STACK) stack32[sp * 64 + lane] = r;
            else vspill32[(size_t)2 * spill_base + (uint32_t)(sp - 2 * STACK)] = r;
        }
        else
        {
            if (sp < STACK) stack[sp][lane] = make_uint2(r, __float_as_uint(entry));
            else
            {
                vspill[(size_t)spill_base + (uint32_t)(sp - STACK)] = make_uint2(r, __float_as_uint(entry));
            }
        }
        ++sp;
    };
    auto pop = [&]()
    {
        ref = RT_IDLE_REF;
        if (SHADOW)
        {
            if (sp > 0)
            {
                --sp;
                if (sp < 2 * STACK) ref = stack32[sp * 64 + lane];
                else ref = spill_load32(vspill32 + (size_t)2 * spill_base + (uint32_t)(sp - 2 * STACK));
            }
        }
        else
            while (sp > 0)
            {
                --sp;
                uint32_t ex, ey;
                if (sp < STACK) { const uint2 e = stack[sp][lane]; ex = e.x; ey = e.y; }
                else { const uint2 e = spill_load64(vspill + (size_t)spill_base + (uint32_t)(sp - STACK)); ex = e.x; ey = e.y; }
                if (t_max >= __uint_as_float(ey)) { ref = ex; break; }       // conservative entry distance: pre-cull only
            }
    };

    // One step of a lane that stands at a leaf (loop B, loop D): `q` = the 64-byte record of triangle `prim`
    auto leaf_step = [&](const float4 q0, const float4 q1, const float4 q2, const float4 q3, const uint32_t prim)
    {
        if (TIMELINE) ++tl_steps;
        bool inside = true;
        if (!(ref & RT_LEAF_CONT_BIT))
        {
            // the reference's RayBounds on the leaf node (trace_bvh.cl:146-148) with the current t_max
            float entry;
            inside = box_test_fast(q1.w, q2.w, q3.x, q3.y, q3.z, q3.w, org, inv, t_min, t_max, entry);
        }
        if (!inside) pop();
        else
        {
            const bool last = q0.w != 0.0f;
            bool accepted = false;
            f3 p1 = F3(q0.x, q0.y, q0.z), e1 = F3(q1.x, q1.y, q1.z), e2 = F3(q2.x, q2.y, q2.z);
            f3 pvec = cross3(dir, e2);
            float det = dot3(e1, pvec);
            if (!(det < 1e-8f || -det > 1e-8f))
            {
                float inv_det = 1.0f / det;
                f3 tvec = org - p1;
                float u = dot3(tvec, pvec) * inv_det;
                if (!(u < 0.0f || u > 1.0f))
                {
                    f3 qvec = cross3(tvec, e1);
                    float v = dot3(dir, qvec) * inv_det;
                    if (!(v < 0.0f || u + v > 1.0f))
                    {
                        float t = dot3(e2, qvec) * inv_det;
                        if (!(t < t_min || t > t_max))
                        {
                            hit_u = u; hit_v = v; hit_prim = prim;
                            t_max = t;                       // :162
                            accepted = true;
                        }
                    }
                }
            }
            if (SHADOW && accepted) ref = RT_IDLE_REF;       // goto endtrace, :164-167
            else if (last) pop();
            else ref = (RT_LEAF_BIT | RT_LEAF_CONT_BIT) | (prim + 1u);
        }
    };
    // One step of a lane that stands at a wide node (loop C, loop D): `q` = its 64-byte record
    auto node_step = [&](const float4 q0, const float4 q1, const float4 q2, const float4 q3)
    {
        if (TIMELINE) ++tl_steps;
        uint32_t r[4];
        float e[4];
        w4_test_slots<SHADOW>(q0, q1, q2, q3, org, inv, sign_bits, octant4, t_min, t_max, r, e);
        // the first passing position is visited next, the later ones wait on the stack (deepest first)
        const bool v0 = e[0] < INF, v1 = e[1] < INF, v2 = e[2] < INF, v3 = e[3] < INF;
        if (v3 && (v0 || v1 || v2)) push(r[3], e[3]);
        if (v2 && (v0 || v1)) push(r[2], e[2]);
        if (v1 && v0) push(r[1], e[1]);
        if (v0) ref = r[0];
        else if (v1) ref = r[1];
        else if (v2) ref = r[2];
        else if (v3) ref = r[3];
        else pop();
    };

    for (;;)
    {
        // ---- A: retire finished rays, start new ones -------------------------------------
        const unsigned long long idle_m = __ballot(ref == RT_IDLE_REF);
        if ((chunk_mode && !(TAIL && chunk_refill)) ? idle_m == ~0ull : idle_m != 0ull)
        {
            if (ref == RT_IDLE_REF && ray_i != RT_INVALID_ID)
            {
                if (SHADOW)
                {
                    if (hit_prim != RT_INVALID_ID)
                        log_retract(log, log_entry, payload);
                }
                else
                    q_store(hits + ray_i, make_float4(hit_u, hit_v, __uint_as_float(hit_prim), t_max));
                if (TIMELINE)
                {
                    const unsigned long long dt = wall_clock64() - tl_t0;
                    if (dt > tl_max_ticks) { tl_max_ticks = dt; tl_steps_of_slowest = tl_steps; }
                    tl_max_steps = tl_steps > tl_max_steps ? tl_steps : tl_max_steps;
                }
                ray_i = RT_INVALID_ID;
            }
            if (!pool.exhausted)
            {
                if (PRIVATE)
                {
                    // the wave's own queue, in order: positions chunk_next .. count - 1 go to the idle lanes
                    const unsigned long long need = __ballot(ref == RT_IDLE_REF && ray_i == RT_INVALID_ID);
                    const uint32_t avail = count - chunk_next;
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
                    const uint32_t n = (uint32_t)__popcll(need);
                    if (ref == RT_IDLE_REF && ray_i == RT_INVALID_ID && rank < avail)
                    {
                        const uint32_t p = chunk_next + rank;
                        ray_i = (priv_first + (p >> 6) * priv_stride) * 64u + (p & 63u);
                    }
                    chunk_next += n < avail ? n : avail;
                    if (chunk_next >= count) pool.exhausted = true;
                }
                else if (TAIL && chunk_mode && chunk_refill)
                {
                    // STATIC chunks, REFILLED lanes (round 4; in the TAIL instance only: in the plain one the extra hand-out path costs the hot
                    // loops 2 % through their register allocation, profiles/r04_call18.log): the wave's chunks -- slot, slot + waves per XCD, ... of its XCD's
                    // region, as below -- are its private queue, and a lane that has finished takes the next ray of it at once
                    // instead of idling until the slowest of its 64 is done.  No atomics (what made refilling from the shared
                    // heads unusable for small launches: one same-address atomic per 64 rays), and each lane still traces a
                    // handful of rays, so the drain at the end is one wave's last rays, not a machine-wide tail.
                    const uint32_t rb = xcd * per < count ? xcd * per : count;
                    const uint32_t re = (xcd + 1u) * per < count ? (xcd + 1u) * per : count;
                    unsigned long long need = __ballot(ref == RT_IDLE_REF && ray_i == RT_INVALID_ID);
                    while (need != 0ull)
                    {
                        if (pool.next >= pool.end)
                        {
                            const uint32_t at = rb + chunk_next * 64u;
                            if (at >= re) { pool.exhausted = true; break; }
                            pool.next = at;
                            pool.end = at + 64u < re ? at + 64u : re;
                            chunk_next += n_blocks >> 3;
                        }
                        const uint32_t avail = pool.end - pool.next;
                        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
                        const uint32_t n = (uint32_t)__popcll(need);
                        if (ref == RT_IDLE_REF && ray_i == RT_INVALID_ID && rank < avail) ray_i = pool.next + rank;
                        pool.next += n < avail ? n : avail;
                        need = __ballot(ref == RT_IDLE_REF && ray_i == RT_INVALID_ID);
                    }
                }
                else if (chunk_mode)
                {
                    // static assignment, no atomics: wave `slot` of this XCD takes chunks slot, slot + waves per XCD, ... of
                    // the XCD's region (64 rays per hand-out through the shared heads costs one same-address atomic per 64
                    // rays -- ~11 M of those per second machine-wide: profiles/r02_handout_sweep.log, r03_call03_*)
                    const uint32_t rb = xcd * per < count ? xcd * per : count;
                    const uint32_t re = (xcd + 1u) * per < count ? (xcd + 1u) * per : count;
                    const uint32_t at = rb + chunk_next * 64u;               // chunk_next: wave-uniform
                    if (at >= re) pool.exhausted = true;
                    else if (at + lane < re) ray_i = at + lane;
                    chunk_next += n_blocks >> 3;
                }
                else
                    hand_out_rays(pool, lane, xcd, per, count, heads, ref == RT_IDLE_REF, ray_i, grab);
                bool slow = false;
                if (ref == RT_IDLE_REF && ray_i != RT_INVALID_ID)
                {
                    const float4 q0 = q_load(o4 + ray_i), q1 = q_load(d4 + ray_i);
                    if (SHADOW) { payload = __float_as_uint(q1.w); log_entry = aux[ray_i]; }
                    org = F3(q0.x, q0.y, q0.z);
                    dir = F3(q1.x, q1.y, q1.z);
                    t_max = q0.w;
                    const float4 q2 = ray_inverse(dir);                      // trace_bvh.cl:125-129
                    inv = F3(q2.x, q2.y, q2.z);
                    sign_bits = __float_as_uint(q2.w) & 0xFFu;
                    octant4 = 4u * (sign_bits & 7u);
                    hit_prim = RT_INVALID_ID;
                    hit_u = 0.0f; hit_v = 0.0f;
                    sp = 0;
                    if (TIMELINE) { tl_steps = 0; tl_t0 = wall_clock64(); }
                    // RT_SIGN_SLOW, or an origin so far out that the one-fma slab distances of loop C could overflow
                    slow = (sign_bits & RT_SIGN_SLOW) != 0u ||
                           !(hw_max3(__builtin_fabsf(org.x), __builtin_fabsf(org.y), __builtin_fabsf(org.z)) < 0x1p29f);
                    if (!slow) ref = entry_ref;
                }
                // rays this kernel does not take: hand their queue index to the BVH2 kernel's list
                const unsigned long long slow_m = __ballot(slow);
                if (slow_m != 0ull)
                {
                    uint32_t base = 0;
                    if (PRIVATE) { base = *slow_count; if (lane == 0) *slow_count = base + (uint32_t)__popcll(slow_m); }     // the wave's own counter: no atomics
                    else
                    {
                        if (lane == 0) { base = atomicAdd(slow_count, (uint32_t)__popcll(slow_m)); atomicAdd(&stat_counts[1], (uint32_t)__popcll(slow_m)); }
                        base = __shfl(base, 0, 64);
                    }
                    if (slow)
                    {
                        uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(slow_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)slow_m, 0u));
                        slow_list[base + rank] = ray_i;
                        ray_i = RT_INVALID_ID;                               // nothing owed: the other kernel writes the result
                    }
                }
            }
        }
        if (TIMELINE && pool.exhausted && !dry_noted)
        {
            dry_noted = true;
            if (lane == 0) atomicMin(&timeline[64], wall_clock64());
        }
        if (__ballot(ref != RT_IDLE_REF) == 0ull)
        {
            if (pool.exhausted) break;                                       // queue dry and every ray retired
            continue;                                                        // a whole wave of rays went to the slow list
        }

        // ---- D: the tail ---------------------------------------------------------------------
        // Loops B and C serve ONE kind of lane per pass, which is what makes a pass cheap while most lanes of the wave have work.
        // When few are left and none can be refilled (the queue is dry, or this is a chunk that must finish before the next
        // starts), a pass is a memory round trip for whoever it serves and a wasted one for the others -- and the launch ends
        // when its LONGEST ray does: ~190 dependent steps on 100 M-ray launches, the 0.3 ms floor of every launch of the
        // reference's one-sample-per-frame pattern (DESIGN.md "Where a launch's time goes").  Here every busy lane fetches its
        // next 64-byte record -- wide node or triangle, the same four loads -- and takes its step in the SAME pass: one round
        // trip per step of every ray.  Per lane the sequence of nodes, leaves and t_max is unchanged.
        if (TAIL && tail_q != 0u && ((chunk_mode && !chunk_refill) || pool.exhausted) && (uint32_t)__popcll(__ballot(ref != RT_IDLE_REF)) <= tail_q)
        {
            do
            {
                if (ref != RT_IDLE_REF)
                {
                    const bool at_leaf = (int)ref < -1;
                    const uint32_t prim = ref & ~(RT_LEAF_BIT | RT_LEAF_CONT_BIT);
                    const float4* rp = reinterpret_cast<const float4*>(at_leaf ? tri_base + (size_t)(prim << 6) : node_base + (size_t)(ref << 6));
                    const float4 q0 = rp[0], q1 = rp[1], q2 = rp[2], q3 = rp[3];
                    if (at_leaf) leaf_step(q0, q1, q2, q3, prim);
                    else node_step(q0, q1, q2, q3);
                }
                n_spills += (uint32_t)__popcll(__ballot(sp > (SHADOW ? 2 * STACK : STACK)));
            } while (__ballot(ref != RT_IDLE_REF) != 0ull);
            continue;                                                        // phase A retires everybody
        }

        // ---- B: leaves: exact box re-test on arrival, then one triangle per pass ----------
        {
            unsigned long long leaf_m = __ballot((int)ref < -1);
            const uint32_t n_node = (uint32_t)__popcll(__ballot((int)ref >= 0));
            if (leaf_m != 0ull && ((uint32_t)__popcll(leaf_m) >= leaf_q || n_node < node_q))
                do
                {
                    if ((int)ref < -1)
                    {
                        const uint32_t prim = ref & ~(RT_LEAF_BIT | RT_LEAF_CONT_BIT);
                        const float4* tp = reinterpret_cast<const float4*>(tri_base + (size_t)(prim << 6));
                        const float4 q0 = tp[0], q1 = tp[1], q2 = tp[2], q3 = tp[3];
                        leaf_step(q0, q1, q2, q3, prim);
                    }
                    leaf_m = __ballot((int)ref < -1);
                } while ((uint32_t)__popcll(leaf_m) >= leaf_q && leaf_m != 0ull);
        }

        // ---- C: wide nodes, the hot loop ----------------------------------------------------
        for (;;)
        {
            const unsigned long long node_m = __ballot((int)ref >= 0);
            if (node_m == 0ull) break;
            if ((uint32_t)__popcll(node_m) < node_q)
            {
                const unsigned long long waiting = __ballot((int)ref < -1 || (ref == RT_IDLE_REF && !pool.exhausted && (!chunk_mode || (TAIL && chunk_refill))));
                if (waiting != 0ull) break;
            }
            if ((int)ref >= 0)
            {
                const float4* np = reinterpret_cast<const float4*>(node_base + (size_t)(ref << 6));
                const float4 q0 = np[0], q1 = np[1], q2 = np[2], q3 = np[3];
                node_step(q0, q1, q2, q3);
            }
            n_spills += (uint32_t)__popcll(__ballot(sp > (SHADOW ? 2 * STACK : STACK)));
        }
    }
    if (lane == 0 && n_spills != 0u) { if (PRIVATE) stat_counts[0] += n_spills; else atomicAdd(&stat_counts[0], n_spills); }
    if (TIMELINE)
    {
        if (lane == 0)
        {
            const unsigned long long now = wall_clock64();
            atomicMax(&timeline[128], now);
            const unsigned long long dry = __atomic_load_n(&timeline[64], __ATOMIC_RELAXED);
            const unsigned long long bin = now > dry ? (now - dry) / 2500ull : 0ull;
            atomicAdd(&timeline[320ull - timeline_slot + (bin < 63ull ? bin : 63ull)], 1ull);   // DCounters::tl_exit_hist
        }
        atomicMax(&timeline[192], (unsigned long long)tl_max_steps);
        atomicMax(&timeline[256], (tl_max_ticks << 24) | (tl_steps_of_slowest & 0xFFFFFFull));
    }
}

template <bool SHADOW, int STACK, bool TIMELINE = false, bool TAIL = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(TAIL ? 7 : 4, 8))) void k_trace_w4(DScene sc, const float4* __restrict__ o4, const float4* __restrict__ d4,
    const uint32_t* __restrict__ aux, const uint32_t* __restrict__ count_ptr, uint32_t* __restrict__ heads,
    float4* __restrict__ hits, DLog log, uint2* __restrict__ spill, uint32_t tune,
    uint32_t* __restrict__ slow_list, uint32_t* __restrict__ slow_count, uint32_t* __restrict__ stat_counts,
    unsigned long long* __restrict__ timeline, uint32_t timeline_slot, uint32_t chunk_below, uint32_t tail_q, uint32_t chunk_refill)
{
    __shared__ uint2 stack[STACK][64];
    w4_trace_body<SHADOW, STACK, TIMELINE, TAIL, false>(sc, o4, d4, aux, *count_ptr, heads, hits, log, spill, tune, slow_list, slow_count, stat_counts, timeline,
        timeline_slot, chunk_below, tail_q, chunk_refill, stack, blockIdx.x, gridDim.x, 0u, 0u);
}
