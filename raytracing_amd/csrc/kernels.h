// kernels.h -- the wavefront path-tracing kernels, hand-written for gfx950.
//
// Reference kernels replaced (src/kernels/cl/):
//   raygeneration.cl:65-139                        -> k_raygen
//   trace_bvh.cl:99-211                            -> k_trace<false>
//   trace_bvh.cl (-D SHADOW_RAYS) + accumulate_direct_samples.cl:27-53
//                                                  -> k_trace<true>
//   miss.cl:41-77 + hit_surface.cl:30-186 + clear_counter.cl (x2)
//                                                  -> k_shade
//   resolve_radiance.cl:31-86                      -> k_resolve
//   reset_radiance.cl / increment_counter.cl       -> hipMemsetAsync / host scalar
//
// Device data layout (HBM), chosen for coalesced 16-byte accesses:
//   ray queues   SoA: o4[i] = (origin.xyz, t_max), d4[i] = (dir.xyz, path id bits),
//                iv4[i] = (1/dir.xyz, sign bits) computed by the PRODUCER of the ray (all 64
//                lanes busy) instead of by the traversal kernel (where ~2 lanes of a wave
//                start a ray in any given iteration), thr[i] = (throughput.xyz, -).  path id = slot * n_pixels + pixel, where
//                `slot` numbers the samples in flight (RT_OPT_SAMPLES_IN_FLIGHT); shadow
//                rays carry (log entry << 25 | path id)
//   radiance log per path: cnt[id] + log[k][id] (float4) = the path's radiance
//                contributions in the order the reference adds them (miss or emission,
//                then direct light, per bounce).  k_flush replays them pixel by pixel,
//                sample by sample, so the fp32 sum is associated exactly as in the
//                reference although several samples are traced concurrently.
//   BVH          one 64-byte "child-pair" record per INTERIOR node of the
//                reference BVH2: both children's boxes + refs in one line, so
//                one dependent fetch serves two box tests (the reference needs
//                one 48-byte fetch per box).  Topology, near/far rule and
//                cull decisions are exactly the reference's (see k_trace).
//   trace tris   64 B, line aligned: (p1, last-in-leaf flag), e1 = p2-p1, e2 = p3-p1, spare
//   shade tris   128 B, line aligned: p1..p3, n1..n3, uv1..uv3, material
#pragma once
#include "device_math.h"
#include "rt_types.h"

#define RT_LEAF_BIT 0x80000000u
#define RT_EMPTY_REF 0xFFFFFFFFu
#define RT_TRACE_STACK_LDS 24     // per-lane stack entries kept in LDS
#define RT_TRACE_STACK_MAX 64     // the reference's nodesToVisit[64] (trace_bvh.cl:142)
// A shadow ray carries its path id in direction.w and, in the w of its 1/direction record,
// sign bits | RT_SIGN_SLOW | (radiance-log entry of its deferred direct sample << 8).

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
    int env_w, env_h;
    uint32_t light_count;
    uint32_t root_ref;            // RT_LEAF_BIT | first triangle, or interior node 0
    uint32_t entry_ref;           // "super-root" record: child 0 = (root box, root_ref), child 1 empty
    float root_min[3];
    float root_max[3];
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

struct DCounters                  // one per frame, device memory
{
    uint32_t queue[64];           // queue[b]  = rays in the incoming queue of bounce b
    uint32_t shadow[64];          // shadow[b] = shadow rays emitted at bounce b
    unsigned long long total_closest, total_shadow, samples;
    uint32_t last_queue[64], last_shadow[64];
    // work-distribution heads of the persistent trace kernels: one per XCD and per
    // kernel flavour (0 = closest, 1 = shadow), offsets inside the XCD's region
    uint32_t head[2][8];
};

// ray_inv_dir and ray_sign of TraceBvh (trace_bvh.cl:125-129), packed as (inv.xyz, sign bits)
#define RT_SIGN_SLOW 8u
RT_DEV float4 ray_inverse(f3 dir)
{
    f3 inv = F3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
    uint32_t sign_bits = (inv.x < 0.0f ? 1u : 0u) | (inv.y < 0.0f ? 2u : 0u) | (inv.z < 0.0f ? 4u : 0u);
    // a non-finite component (dir component 0, denormal or NaN) can make the slab test produce
    // 0 * inf = NaN: such rays keep the select-form min/max of the reference (box_test)
    const float inf = __builtin_inff();
    if (!(__builtin_fabsf(inv.x) < inf && __builtin_fabsf(inv.y) < inf && __builtin_fabsf(inv.z) < inf))
        sign_bits |= RT_SIGN_SLOW;
    return make_float4(inv.x, inv.y, inv.z, __uint_as_float(sign_bits));
}

// ---------------------------------------------------------------------------
// sample begin + ray generation (raygeneration.cl:65-139)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_raygen(DTile tile, rt_camera cam, uint32_t sample_base, uint32_t n_slots,
    float tan_half_fov, uint32_t prev_bounces, float4* __restrict__ o4, float4* __restrict__ d4,
    float4* __restrict__ iv4, float4* __restrict__ thr, DCounters* __restrict__ counters)
{
    uint32_t n_local = tile.local_rows * tile.width;
    uint32_t n_total = n_local * n_slots;                                // n_slots samples in flight
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
            counters->last_queue[b] = counters->queue[b];
            counters->last_shadow[b] = counters->shadow[b];
        }
        counters->total_closest += c;
        counters->total_shadow += s;
        for (uint32_t b = 0; b < 64; ++b) { counters->queue[b] = 0; counters->shadow[b] = 0; }
        counters->queue[0] = n_total;                                    // raygeneration.cl:135-138
    }
    if (i < 16) counters->head[i >> 3][i & 7] = 0;                       // the next trace launches start from 0
    if (i >= n_total) return;

    // queue order is PIXEL-major: the n_slots samples of a pixel sit next to each other, so
    // a wave holds a few pixels x all their samples -- nearly identical primary rays, and
    // secondary/shadow rays that start from the same small surface patch (shadow rays towards
    // a directional light are then almost parallel AND co-located).  Fewer distinct BVH
    // records per load instruction is what the L1 data path rewards.  The path id keeps the
    // slot-major form (slot * n_local + pixel) the radiance log is laid out by.
    uint32_t lp = i / n_slots;                                           // local pixel of this tile
    uint32_t slot = i - lp * n_slots;
    uint32_t sample_idx = sample_base + slot;
    uint32_t ly = lp / tile.width;
    uint32_t pixel_x = lp - ly * tile.width;
    uint32_t pixel_y = tile_global_row(tile, ly);
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
    f3 new_pos = pos + right * (dofx * r) + up * (dofy * r);
    f3 d = normalize3(point_aimed - new_pos);

    o4[i] = make_float4(new_pos.x, new_pos.y, new_pos.z, RT_MAX_RENDER_DIST);
    d4[i] = make_float4(d.x, d.y, d.z, __uint_as_float(slot * n_local + lp));   // path id
    iv4[i] = ray_inverse(d);
    thr[i] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
}

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

template <bool SHADOW>
__global__ __launch_bounds__(64) void k_trace_v1(DScene sc, const float4* __restrict__ o4, const float4* __restrict__ d4,
    const float4* __restrict__ iv4, const uint32_t* __restrict__ count_ptr, float4* __restrict__ hits,
    float4* __restrict__ rlog, uint32_t log_stride, uint32_t /*force_sign_bits: v1 always uses box_test*/,
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

    for (uint32_t c = slot; c < cpx; c += per_xcd)
    {
        uint32_t chunk = xcd * cpx + c;
        uint32_t i = chunk * 64u + lane;
        if (i >= count) continue;

        float4 ro = o4[i], rd = d4[i];
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
                bool h0 = box_test(n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, org, inv, t_min, t_max, a0);
                bool h1 = box_test(n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, org, inv, t_min, t_max, a1);
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
                        uint2 e = make_uint2(far_ref, __float_as_uint(far_entry));
                        if (sp < RT_TRACE_STACK_LDS) stack[sp][lane] = e;
                        else my_spill[sp - RT_TRACE_STACK_LDS] = e;
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
                    uint2 e = (sp < RT_TRACE_STACK_LDS) ? stack[sp][lane] : my_spill[sp - RT_TRACE_STACK_LDS];
                    if (t_max >= __uint_as_float(e.y))                   // box re-test at pop time
                    {
                        ref = e.x;
                        alive = true;
                        break;
                    }
                }
            }
        }

        if (SHADOW)
        {
            // AccumulateDirectSamples (accumulate_direct_samples.cl:46-52) fused: k_shade
            // logged the direct sample tentatively; an occluded ray retracts it
            if (occluded)
            {
                uint32_t entry = __float_as_uint(iv4[i].w) >> 8;
                rlog[(size_t)entry * log_stride + __float_as_uint(rd.w)] = make_float4(0, 0, 0, 0);
            }
        }
        else
        {
            hits[i] = make_float4(hit_u, hit_v, __uint_as_float(hit_prim), hit_t);
        }
    }
}


// ---------------------------------------------------------------------------
// k_trace: persistent "one fetch per iteration" traversal (the production kernel)
// ---------------------------------------------------------------------------
// What bounds this kernel is the chain of dependent HBM/L2 round trips per ray
// (~46 box tests + ~2.5 triangle tests per ray on the 890 k-triangle stand-in),
// not arithmetic.  v1 above pays (a) one round trip for the node branch PLUS one
// for the leaf branch whenever a wave has lanes in both, and (b) idles lanes
// whose ray finished until the slowest ray of the wave is done.  Here every lane
// is a small state machine and every loop iteration issues exactly ONE 64-byte
// record fetch per lane -- the next ray (o4/d4), a child-pair node, or a
// triangle -- through the same four load instructions, so the wave pays one
// memory round trip per iteration whatever mix of states it holds, and a lane
// that finishes pulls a new ray in the very next iteration (wave-level pool of
// ray indices, refilled RT_TRACE_BATCH at a time from a per-XCD queue head with
// one atomic; exhausted XCD regions steal from the next region).
// The arithmetic per record is unchanged from v1 (bit-identical results).
#define RT_TRACE_BATCH 128u
enum { ST_NEED = 0, ST_RAY = 1, ST_TRAV = 2, ST_DONE = 3 };

template <bool SHADOW, int STACK>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(STACK <= 10 ? 8 : 6, 8))) void k_trace(DScene sc, const float4* __restrict__ o4, const float4* __restrict__ d4,
    const float4* __restrict__ iv4, const uint32_t* __restrict__ count_ptr, uint32_t* __restrict__ heads,
    float4* __restrict__ hits,
    float4* __restrict__ rlog, uint32_t log_stride, uint32_t force_sign_bits, uint2* __restrict__ spill)
{
    __shared__ uint2 stack[STACK][64];
    const uint32_t lane = threadIdx.x;
    const uint32_t count = *count_ptr;
    if (count == 0) return;
    const uint32_t xcd = blockIdx.x & 7u;
    // eight contiguous regions of the queue, 64-ray aligned, one per XCD (L2 affinity)
    const uint32_t per = (((count + 7u) >> 3) + 63u) & ~63u;
    uint2* my_spill = spill + (size_t)(blockIdx.x * 64u + lane) * (RT_TRACE_STACK_MAX - STACK);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    uint32_t pool_next = 0, pool_end = 0, regions_tried = 0;   // wave-uniform
    uint32_t state = ST_NEED;
    uint32_t ray_i = 0, ref = 0, sign_bits = 0, hit_prim = RT_INVALID_ID;
    int sp = 0;
    f3 org = F3s(0.0f), dir = F3s(0.0f), inv = F3s(0.0f);
    float t_max = 0.0f, hit_u = 0.0f, hit_v = 0.0f;
    uint32_t payload = 0, log_entry = 0;                                     // SHADOW: path id, radiance-log entry
    const float t_min = 0.0f;

    for (;;)
    {
        // ---- hand new ray indices to the lanes that need one --------------------
        unsigned long long need = __ballot(state == ST_NEED);
        while (need)
        {
            if (pool_next >= pool_end)
            {
                bool got = false;
                while (regions_tried < 8u)
                {
                    uint32_t x = (xcd + regions_tried) & 7u;
                    uint32_t rb = x * per < count ? x * per : count;
                    uint32_t re = (x + 1u) * per < count ? (x + 1u) * per : count;
                    uint32_t b = 0;
                    if (lane == 0 && rb < re) b = atomicAdd(&heads[x], RT_TRACE_BATCH);
                    b = __shfl(b, 0, 64);
                    if (rb < re && b < re - rb)
                    {
                        pool_next = rb + b;
                        pool_end = (b + RT_TRACE_BATCH < re - rb) ? rb + b + RT_TRACE_BATCH : re;
                        got = true;
                        break;
                    }
                    ++regions_tried;
                }
                if (!got)
                {
                    if (state == ST_NEED) state = ST_DONE;
                    break;
                }
            }
            uint32_t avail = pool_end - pool_next;
            uint32_t rank = (uint32_t)__popcll(need & lt_mask);
            uint32_t n = (uint32_t)__popcll(need);
            if (state == ST_NEED && rank < avail)
            {
                ray_i = pool_next + rank;
                state = ST_RAY;
            }
            pool_next += n < avail ? n : avail;
            need = __ballot(state == ST_NEED);
        }
        if (__ballot(state != ST_DONE) == 0ull) break;

        // ---- ONE 64-byte record per lane: ray | node | triangle -----------------
        // (the L1 sees one access per lane per load instruction: only child-pair nodes use
        // the fourth 16 bytes, so triangle and ray lanes skip that load)
        const float4 *p0, *p1, *p2;
        const bool is_node = state == ST_TRAV && !(ref & RT_LEAF_BIT);
        if (state == ST_RAY) { p0 = o4 + ray_i; p1 = d4 + ray_i; p2 = iv4 + ray_i; }
        else
        {
            const float4* base = (ref & RT_LEAF_BIT) ? sc.tris_rt + (size_t)(ref & ~RT_LEAF_BIT) * 4
                                                     : sc.nodes + (size_t)ref * 4;
            p0 = base; p1 = base + 1; p2 = base + 2;
        }
        float4 q0, q1, q2, q3;
        if (state != ST_DONE)
        {
            q0 = *p0; q1 = *p1; q2 = *p2;
            if (is_node) q3 = p2[1];
        }
        if (SHADOW && state == ST_RAY) { payload = __float_as_uint(q1.w); log_entry = __float_as_uint(q2.w) >> 8; }

        bool finished = false, need_pop = false;
        if (state == ST_RAY)
        {
            // ray start: registers only.  1/dir and the sign bits come from the producer
            // (ray_inverse); the root box test (trace_bvh.cl:146-148, first iteration) is the
            // ordinary node test of the "super-root" record entry_ref in the next iteration.
            org = F3(q0.x, q0.y, q0.z);
            dir = F3(q1.x, q1.y, q1.z);
            t_max = q0.w;
            inv = F3(q2.x, q2.y, q2.z);
            sign_bits = (__float_as_uint(q2.w) & 0xFFu) | force_sign_bits;
            hit_prim = RT_INVALID_ID;
            hit_u = 0.0f; hit_v = 0.0f;
            sp = 0;
            ref = sc.entry_ref;
            state = ST_TRAV;
        }
        else if (state == ST_TRAV)
        {
            if (ref & RT_LEAF_BIT)
            {
                // one triangle of a leaf (trace_bvh.cl:28-73,155-169)
                uint32_t prim = ref & ~RT_LEAF_BIT;
                bool last = q0.w != 0.0f;
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
                                t_max = t;                                   // :162
                                if (SHADOW) finished = true;                 // goto endtrace, :164-167
                            }
                        }
                    }
                }
                if (!finished)
                {
                    if (last) need_pop = true;
                    else ref = ref + 1u;
                }
            }
            else
            {
                uint32_t c0 = __float_as_uint(q3.x), c1 = __float_as_uint(q3.y), axis = __float_as_uint(q3.z);
                float a0, a1;
                bool h0, h1;
                if (sign_bits & RT_SIGN_SLOW)
                {
                    h0 = box_test(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, org, inv, t_min, t_max, a0);
                    h1 = box_test(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, org, inv, t_min, t_max, a1);
                }
                else
                {
                    h0 = box_test_fast(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, org, inv, t_min, t_max, a0);
                    h1 = box_test_fast(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, org, inv, t_min, t_max, a1);
                }
                h1 = h1 && (c1 != RT_EMPTY_REF);
                bool swap = (sign_bits >> axis) & 1u;                        // :181-190
                uint32_t near_ref = swap ? c1 : c0, far_ref = swap ? c0 : c1;
                bool near_hit = swap ? h1 : h0, far_hit = swap ? h0 : h1;
                float far_entry = swap ? a0 : a1;
                if (near_hit)
                {
                    if (far_hit)
                    {
                        uint2 e = make_uint2(far_ref, __float_as_uint(far_entry));
                        if (sp < STACK) stack[sp][lane] = e;
                        else my_spill[sp - STACK] = e;
                        ++sp;
                    }
                    ref = near_ref;
                }
                else if (far_hit) ref = far_ref;
                else need_pop = true;
            }
            if (need_pop)
            {
                finished = true;
                while (sp > 0)
                {
                    --sp;
                    uint2 e = (sp < STACK) ? stack[sp][lane] : my_spill[sp - STACK];
                    if (t_max >= __uint_as_float(e.y))                       // box re-test at pop time
                    {
                        ref = e.x;
                        finished = false;
                        break;
                    }
                }
            }
        }

        if (finished)
        {
            if (SHADOW)
            {
                // AccumulateDirectSamples fused (accumulate_direct_samples.cl:46-52): k_shade
                // logged the direct sample tentatively; an occluded ray (it stopped on its
                // first accepted triangle, hit_prim set) retracts it.  Store only, no wait.
                if (hit_prim != RT_INVALID_ID)
                    rlog[(size_t)log_entry * log_stride + payload] = make_float4(0, 0, 0, 0);
            }
            else
            {
                hits[ray_i] = make_float4(hit_u, hit_v, __uint_as_float(hit_prim), t_max);
            }
            state = ST_NEED;
        }
    }
}

// ---------------------------------------------------------------------------
// Packet traversal for COHERENT ray batches (primary rays and the shadow rays of the first
// hit): the 64 consecutive queue entries of a wave -- samples of one pixel, or neighbouring
// pixels -- walk the tree TOGETHER.  The current node is wave-uniform, so its 64-byte record
// is fetched once per wave by the scalar unit (s_load_dwordx16 through the scalar cache: no
// vector-memory instruction, no L1 access, no per-lane address arithmetic) and the box /
// triangle tests read it from SGPRs; the stack holds wave-uniform entries.
//
// Exactness: lanes are grouped by their three direction sign bits, so every lane of a group
// orders children exactly as the reference does for its ray (trace_bvh.cl:181-190).  A lane
// takes part in a node visit iff its OWN box test of that node passes with its OWN current
// t_max -- the reference's pop-time test (trace_bvh.cl:146-148): for the near child that is
// the test made while the parent is visited (nothing happens to the ray in between); a far
// child is pushed as (parent, child index, lanes that hit the parent) when any lane hits it
// now (a lane that misses it now misses it later, t_max only shrinks), and when it is popped
// the parent record is fetched again and those lanes re-run the full box test with their
// current t_max.  Each lane therefore sees exactly its reference sequence of nodes and
// triangles -- a subsequence of the packet's -- and produces the same hit.
RT_DEV uint32_t uniform_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

struct Rec64 { float4 a, b, c, d; };
typedef float rt_v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) rt_v4f* const_v4f_ptr;     // constant address space -> s_load
RT_DEV Rec64 scalar_fetch(const float4* base, uint32_t uniform_index)
{
    const_v4f_ptr p = (const_v4f_ptr)(uintptr_t)(base + (size_t)uniform_u32(uniform_index) * 4);
    rt_v4f a = p[0], b = p[1], c = p[2], d = p[3];
    Rec64 r;
    r.a = make_float4(a.x, a.y, a.z, a.w); r.b = make_float4(b.x, b.y, b.z, b.w);
    r.c = make_float4(c.x, c.y, c.z, c.w); r.d = make_float4(d.x, d.y, d.z, d.w);
    return r;
}

RT_DEV bool packet_box(bool slow, float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz, f3 org, f3 inv,
    float t_max)
{
    float entry;
    return slow ? box_test(bminx, bminy, bminz, bmaxx, bmaxy, bmaxz, org, inv, 0.0f, t_max, entry)
                : box_test_fast(bminx, bminy, bminz, bmaxx, bmaxy, bmaxz, org, inv, 0.0f, t_max, entry);
}

template <bool SHADOW>
__global__ __launch_bounds__(64) void k_trace_packet(DScene sc, const float4* __restrict__ o4, const float4* __restrict__ d4,
    const float4* __restrict__ iv4, const uint32_t* __restrict__ count_ptr, uint32_t* __restrict__ heads,
    float4* __restrict__ hits, float4* __restrict__ rlog, uint32_t log_stride, uint32_t force_sign_bits)
{
    __shared__ uint4 pstack[RT_TRACE_STACK_MAX + 1];                         // wave-uniform entries
    const uint32_t lane = threadIdx.x;
    const uint32_t count = *count_ptr;
    if (count == 0) return;
    const uint32_t xcd = blockIdx.x & 7u;
    const uint32_t per = (((count + 7u) >> 3) + 63u) & ~63u;                 // the XCD regions of k_trace
    const unsigned long long lane_bit = 1ull << lane;
    uint32_t regions_tried = 0;

    for (;;)
    {
        // ---- next packet: 64 consecutive queue entries of this XCD's region (then steal) ----
        uint32_t base = 0, end = 0;
        bool got = false;
        while (regions_tried < 8u)
        {
            uint32_t x = (xcd + regions_tried) & 7u;
            uint32_t rb = x * per < count ? x * per : count;
            uint32_t re = (x + 1u) * per < count ? (x + 1u) * per : count;
            uint32_t b = 0;
            if (lane == 0 && rb < re) b = atomicAdd(&heads[x], 64u);
            b = uniform_u32(b);
            if (rb < re && b < re - rb) { base = rb + b; end = re; got = true; break; }
            ++regions_tried;
        }
        if (!got) break;

        const uint32_t i = base + lane;
        const bool valid = i < end;
        f3 org = F3s(0.0f), dir = F3s(0.0f), inv = F3s(0.0f);
        float t_max = 0.0f, hit_u = 0.0f, hit_v = 0.0f;
        uint32_t sign_bits = 0, hit_prim = RT_INVALID_ID, payload = 0, log_entry = 0;
        if (valid)
        {
            float4 q0 = o4[i], q1 = d4[i], q2 = iv4[i];
            org = F3(q0.x, q0.y, q0.z); t_max = q0.w;
            dir = F3(q1.x, q1.y, q1.z); payload = __float_as_uint(q1.w);
            inv = F3(q2.x, q2.y, q2.z);
            sign_bits = (__float_as_uint(q2.w) & 0xFFu) | force_sign_bits;
            log_entry = __float_as_uint(q2.w) >> 8;
        }

        unsigned long long todo = __ballot(valid);
        while (todo)
        {
            // one group = the lanes that share the first pending lane's direction signs
            const uint32_t first = (uint32_t)__ffsll((long long)todo) - 1u;
            const uint32_t sgn = uniform_u32((uint32_t)__shfl((int)(sign_bits & 7u), (int)first, 64));
            const unsigned long long group = __ballot(valid && (sign_bits & 7u) == sgn) & todo;
            todo &= ~group;
            const bool slow = __ballot((group & lane_bit) && (sign_bits & RT_SIGN_SLOW)) != 0ull;   // wave-uniform

            unsigned long long alive = group;          // SHADOW: lanes leave on their first accepted hit
            unsigned long long mask = group;           // lanes taking part in the current visit
            uint32_t ref = sc.entry_ref;
            int sp = 0;
            for (;;)
            {
                bool need_pop = false;
                if (ref & RT_LEAF_BIT)
                {
                    uint32_t prim = ref & ~RT_LEAF_BIT;
                    for (;;)
                    {
                        const Rec64 t = scalar_fetch(sc.tris_rt, prim);
                        const bool last = t.a.w != 0.0f;
                        bool accepted = false;
                        if (mask & alive & lane_bit)
                        {
                            // RayTriangle, trace_bvh.cl:28-73,155-169
                            f3 p1 = F3(t.a.x, t.a.y, t.a.z), e1 = F3(t.b.x, t.b.y, t.b.z), e2 = F3(t.c.x, t.c.y, t.c.z);
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
                                        float tt = dot3(e2, qvec) * inv_det;
                                        if (!(tt < 0.0f || tt > t_max))
                                        {
                                            hit_u = u; hit_v = v; hit_prim = prim;
                                            t_max = tt;
                                            accepted = true;
                                        }
                                    }
                                }
                            }
                        }
                        if (SHADOW) alive &= ~__ballot(accepted);            // goto endtrace, :164-167
                        if (last || (mask & alive) == 0ull) break;
                        ++prim;
                    }
                    need_pop = true;
                }
                else
                {
                    const Rec64 n = scalar_fetch(sc.nodes, ref);
                    const uint32_t c0 = __float_as_uint(n.d.x), c1 = __float_as_uint(n.d.y), axis = __float_as_uint(n.d.z);
                    bool h0 = false, h1 = false;
                    if (mask & alive & lane_bit)
                    {
                        h0 = packet_box(slow, n.a.x, n.a.y, n.a.z, n.a.w, n.b.x, n.b.y, org, inv, t_max);
                        h1 = packet_box(slow, n.b.z, n.b.w, n.c.x, n.c.y, n.c.z, n.c.w, org, inv, t_max);
                    }
                    const unsigned long long m0 = __ballot(h0);
                    const unsigned long long m1 = c1 != RT_EMPTY_REF ? __ballot(h1) : 0ull;
                    const bool swap = (sgn >> axis) & 1u;                    // :181-190, the same for the whole group
                    const unsigned long long m_near = swap ? m1 : m0, m_far = swap ? m0 : m1;
                    const uint32_t near_ref = swap ? c1 : c0, far_ref = swap ? c0 : c1;
                    if (m_near)
                    {
                        if (m_far)
                        {
                            const unsigned long long pm = mask & alive;
                            if (lane == 0) pstack[sp] = make_uint4(ref, swap ? 0u : 1u, (uint32_t)pm, (uint32_t)(pm >> 32));
                            ++sp;
                        }
                        ref = near_ref; mask = m_near;
                    }
                    else if (m_far) { ref = far_ref; mask = m_far; }
                    else need_pop = true;
                }
                if (need_pop)
                {
                    bool found = false;
                    while (sp > 0 && alive)
                    {
                        --sp;
                        const uint4 e = pstack[sp];
                        const uint32_t pref = uniform_u32(e.x), cidx = uniform_u32(e.y);
                        const unsigned long long pm =
                            (((unsigned long long)uniform_u32(e.w) << 32) | uniform_u32(e.z)) & alive;
                        if (pm == 0ull) continue;
                        const Rec64 n = scalar_fetch(sc.nodes, pref);
                        bool h = false;
                        if (pm & lane_bit)
                            h = cidx ? packet_box(slow, n.b.z, n.b.w, n.c.x, n.c.y, n.c.z, n.c.w, org, inv, t_max)
                                     : packet_box(slow, n.a.x, n.a.y, n.a.z, n.a.w, n.b.x, n.b.y, org, inv, t_max);
                        const unsigned long long m = __ballot(h);
                        if (m)
                        {
                            ref = cidx ? __float_as_uint(n.d.y) : __float_as_uint(n.d.x);
                            mask = m;
                            found = true;
                            break;
                        }
                    }
                    if (!found) break;
                }
            }
        }

        if (valid)
        {
            if (SHADOW)
            {
                if (hit_prim != RT_INVALID_ID)
                    rlog[(size_t)log_entry * log_stride + payload] = make_float4(0, 0, 0, 0);
            }
            else
            {
                hits[i] = make_float4(hit_u, hit_v, __uint_as_float(hit_prim), t_max);
            }
        }
    }
}

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

RT_DEV f3 pow3(f3 a, float e) { return F3(rt_powf(a.x, e), rt_powf(a.y, e), rt_powf(a.z, e)); }

RT_DEV f3 UnpackRGBTex(uint32_t data, uint32_t& idx)                    // utils.h:133-147
{
    float r = (float)(data & 0xFF), g = (float)((data >> 8) & 0xFF), b = (float)((data >> 16) & 0xFF);
    idx = (data >> 24) & 0xFF;
    return F3(r / 255.0f, g / 255.0f, b / 255.0f);
}

RT_DEV void ApplyTextures(const DScene& sc, rt_packed_material in, Material& out, f2 uv)   // material.h:319-369
{
    uint32_t idx;
    out.diffuse_albedo = UnpackRGBTex(in.diffuse_albedo, idx);
    if (idx != RT_INVALID_TEXTURE_IDX) out.diffuse_albedo = pow3(SampleTexture(sc, idx, uv), 2.2f);
    out.specular_albedo = UnpackRGBTex(in.specular_albedo, idx);
    if (idx != RT_INVALID_TEXTURE_IDX) out.specular_albedo = pow3(SampleTexture(sc, idx, uv), 2.2f);
    {
        uint32_t rgbe = in.emission;                                     // utils.h:149-158
        int r = (int)(rgbe & 0xFF), g = (int)((rgbe >> 8) & 0xFF), b = (int)((rgbe >> 16) & 0xFF);
        int e = (int)(rgbe >> 24);
        float f = rt_ldexpf(1.0f, e - (128 + 8));
        out.emission = F3((float)r * f, (float)g * f, (float)b * f);
    }
    uint32_t d = in.roughness_metalness;                                 // utils.h:160-174
    out.roughness = (float)(d & 0xFF) / 255.0f;
    uint32_t roughness_idx = (d >> 8) & 0xFF;
    out.metalness = (float)((d >> 16) & 0xFF) / 255.0f;
    uint32_t metalness_idx = (d >> 24) & 0xFF;
    if (roughness_idx != RT_INVALID_TEXTURE_IDX) out.roughness = SampleTexture(sc, roughness_idx, uv).x;
    if (metalness_idx != RT_INVALID_TEXTURE_IDX) out.metalness = SampleTexture(sc, metalness_idx, uv).x;
    d = in.ior_emission_idx_transparency;                                // utils.h:176-190
    out.ior = (float)(d & 0xFF) / 25.5f;
    uint32_t emission_idx = (d >> 8) & 0xFF;
    out.transparency = (float)((d >> 16) & 0xFF) / 255.0f;
    uint32_t transparency_idx = (d >> 24) & 0xFF;
    if (emission_idx != RT_INVALID_TEXTURE_IDX)
        out.emission = out.emission * pow3(SampleTexture(sc, emission_idx, uv), 2.2f);
    if (transparency_idx != RT_INVALID_TEXTURE_IDX)
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
    float& offset)
{
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
        return F3s(1.0f);
    }

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
        }
        else
        {
            // GGX_Sample bxdf.h:157-168 (fp64 literals in the reference -> fp64 divide + sqrt)
            float phi = RT_TWO_PI * s.x;
            float cos_theta = (float)(1.0 / __builtin_sqrt(1.0 + (double)(alpha * alpha * s.y) / (1.0 - (double)s.y)));
            float sin_theta = __builtin_sqrtf(cl_max(0.0f, 1.0f - cos_theta * cos_theta));
            f3 t, b;
            tangent_frame(normal, t, b);
            double sd, cd;
            rtd_sincos((double)phi, &sd, &cd);
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
        float phi = RT_TWO_PI * s.x;
        float sin_theta = __builtin_sqrtf(s.y);
        float cos_theta = __builtin_sqrtf(1.0f - s.y);
        pdf = cos_theta * RT_INV_PI;
        double sd, cd;
        rtd_sincos((double)phi, &sd, &cd);
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
    int i0 = (int)fu, j0 = (int)fv;
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
#define RT_SHADE_BLOCK 512
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

struct ShadeArgs
{
    const float4* in_o4; const float4* in_d4; const float4* in_thr; const float4* hits;
    float4* out_o4; float4* out_d4; float4* out_iv4; float4* out_thr;
    float4* sh_o4; float4* sh_d4; float4* sh_iv4;
    float4* rlog; uint32_t* cnt;      // radiance log (see file header)
    const uint8_t* bn_sobol; const uint8_t* bn_scramble; const uint8_t* bn_rank;   // SamplerType::kBlueNoise tables
    DCounters* counters;
    uint32_t bounce, sample_base, emit_outgoing, n_local, log_stride;
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

template <bool FURNACE, bool BLUE>
__global__ __launch_bounds__(RT_SHADE_BLOCK) void k_shade(DScene sc, DTile tile, ShadeArgs a)
{
    const uint32_t count = a.counters->queue[a.bounce];
    const uint32_t i = blockIdx.x * RT_SHADE_BLOCK + threadIdx.x;
    // the closest-hit trace of this bounce has completed (stream order): rewind the
    // work heads for the shadow trace of this bounce and the closest trace of the next
    if (i < 16) a.counters->head[i >> 3][i & 7] = 0;
    if (blockIdx.x * RT_SHADE_BLOCK >= count) return;                    // whole block idle (uniform)
    const bool active = i < count;

    bool want_shadow = false, want_next = false;
    float4 sh_o = make_float4(0, 0, 0, 0), sh_d = sh_o, nx_o = sh_o, nx_d = sh_o, nx_t = sh_o;
    uint32_t sh_entry = 0;

    if (active)
    {
        float4 hit = a.hits[i];
        float4 rd = a.in_d4[i];
        uint32_t prim = __float_as_uint(hit.z);
        uint32_t id = __float_as_uint(rd.w);                               // slot * n_local + local pixel
        uint32_t slot = id / a.n_local;
        uint32_t pix = id - slot * a.n_local;
        uint32_t sample_idx = a.sample_base + slot;
        uint32_t nlog = a.cnt[id];                                         // contributions logged so far
        float4* mylog = a.rlog + id;
        float4 thr4 = a.in_thr[i];
        f3 hit_throughput = F3(thr4.x, thr4.y, thr4.z);

        if (prim == RT_INVALID_ID)
        {
            // Miss, miss.cl:65-76
            f3 sky = FURNACE ? F3s(0.5f) : SampleSky(sc, F3(rd.x, rd.y, rd.z));
            f3 add = sky * hit_throughput;
            mylog[(size_t)nlog * a.log_stride] = make_float4(add.x, add.y, add.z, 0.0f);   // radiance[pix] += ...
            ++nlog;
        }
        else
        {
            // HitSurface, hit_surface.cl:79-184
            f3 incoming = F3(-rd.x, -rd.y, -rd.z);
            uint32_t ly = pix / tile.width;
            uint32_t px = pix - ly * tile.width;
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
            ApplyTextures(sc, sc.materials[__float_as_uint(q6.x)], material, texcoord);

            if (!FURNACE)
            {
                if (material.emission.x * 1.0f + material.emission.y * 1.0f + material.emission.z * 1.0f > 0.0f)
                {
                    f3 e = hit_throughput * material.emission;
                    mylog[(size_t)nlog * a.log_stride] = make_float4(e.x, e.y, e.z, 0.0f);         // radiance[pix] += ...
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
                int light_idx = cl_clampi((int)(s_light * (float)sc.light_count), 0, (int)sc.light_count - 1);
                float4 lo = sc.lights[light_idx * 3 + 0], lr = sc.lights[light_idx * 3 + 1];
                uint32_t ltype = __float_as_uint(sc.lights[light_idx * 3 + 2].x);
                float pdf = 1.0f / (float)sc.light_count;
                f3 light_radiance = xyz(lr);
                f3 outgoing;
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
                    mylog[(size_t)nlog * a.log_stride] = make_float4(lsamp.x, lsamp.y, lsamp.z, 0.0f);
                    ++nlog;
                }
            }

            // Indirect lighting :148-184
            {
                f2 s;
                s.x = draw(2);
                s.y = draw(3);
                float s1 = draw(1);
                float pdf = 0.0f;
                f3 outgoing;
                float offset;
                f3 bxdf = SampleBxdf<FURNACE>(s1, s, material, normal, incoming, outgoing, pdf, offset);
                f3 throughput = F3s(0.0f);
                if ((double)pdf > 0.0) throughput = bxdf / pdf;
                f3 new_thr = hit_throughput * throughput;                 // throughputs[pixel] *= throughput
                want_next = ((double)pdf > 0.0) && (a.emit_outgoing != 0);
                f3 oo = position + geometry_normal * RT_EPS * offset;
                nx_o = make_float4(oo.x, oo.y, oo.z, RT_MAX_RENDER_DIST);
                nx_d = make_float4(outgoing.x, outgoing.y, outgoing.z, rd.w);
                nx_t = make_float4(new_thr.x, new_thr.y, new_thr.z, 0.0f);
            }
        }
        a.cnt[id] = nlog;
    }

    uint32_t sidx, nidx;
    block_append2(want_shadow, want_next, &a.counters->shadow[a.bounce], &a.counters->queue[a.bounce + 1], sidx, nidx);
    if (want_shadow)
    {
        a.sh_o4[sidx] = sh_o;
        a.sh_d4[sidx] = sh_d;
        float4 siv = ray_inverse(F3(sh_d.x, sh_d.y, sh_d.z));
        siv.w = __uint_as_float(__float_as_uint(siv.w) | (sh_entry << 8));
        a.sh_iv4[sidx] = siv;
    }
    if (want_next)
    {
        a.out_o4[nidx] = nx_o;
        a.out_d4[nidx] = nx_d;
        a.out_iv4[nidx] = ray_inverse(F3(nx_d.x, nx_d.y, nx_d.z));
        a.out_thr[nidx] = nx_t;
    }
}

// Replays the radiance log: for every pixel, sample slot by sample slot, contribution
// by contribution -- the exact order in which the reference's kernels executed
// `radiance[pixel] += ...` (miss.cl:75, hit_surface.cl:110, accumulate_direct_samples.cl:51).
__global__ __launch_bounds__(256) void k_flush(float4* __restrict__ radiance, const float4* __restrict__ rlog,
    uint32_t* __restrict__ cnt, uint32_t n_local, uint32_t n_slots, uint32_t log_stride)
{
    uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n_local) return;
    float4 r = radiance[p];
    for (uint32_t slot = 0; slot < n_slots; ++slot)
    {
        uint32_t id = slot * n_local + p;
        uint32_t c = cnt[id];
        for (uint32_t k = 0; k < c; ++k)
        {
            float4 v = rlog[(size_t)k * log_stride + id];
            r.x += v.x; r.y += v.y; r.z += v.z;
        }
        if (c) cnt[id] = 0;
    }
    radiance[p] = r;
}

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
    ApplyTextures(sc, sc.materials[__float_as_uint(q6.x)], material, texcoord);
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

// end-of-run fold of the per-bounce counters (same as the prologue of k_raygen)
__global__ void k_fold_counters(DCounters* counters, uint32_t bounces)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned long long c = 0, s = 0;
    for (uint32_t b = 0; b <= bounces && b < 64; ++b)
    {
        c += counters->queue[b];
        s += counters->shadow[b];
        counters->last_queue[b] = counters->queue[b];
        counters->last_shadow[b] = counters->shadow[b];
        counters->queue[b] = 0;
        counters->shadow[b] = 0;
    }
    counters->total_closest += c;
    counters->total_shadow += s;
}

// ---------------------------------------------------------------------------
// Scene re-layout on the device (rt_scene_upload): the reference's arrays are copied to HBM
// as they are and three streaming kernels write the traversal / shading layouts.
// Error codes (first one wins) are decoded by the host.
// ---------------------------------------------------------------------------
enum { RL_OK = 0, RL_CHILD_RANGE = 1, RL_LEAF_RANGE = 2, RL_AXIS = 3, RL_MATERIAL = 4 };

// one thread per LinearBVHNode: leaves mark their last triangle
__global__ void k_relayout_mark_leaves(const rt_bvh_node* __restrict__ nodes, uint32_t nn, uint32_t nt,
    uint8_t* __restrict__ last_in_leaf, int* __restrict__ err)
{
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nn) return;
    uint32_t n = nodes[i].num_primitives_axis >> 16;
    if (n == 0) return;
    uint32_t first = nodes[i].offset;
    if ((unsigned long long)first + n > nt) { atomicCAS(err, RL_OK, RL_LEAF_RANGE); return; }
    last_in_leaf[first + n - 1u] = 1;
}

// one thread per LinearBVHNode: interior nodes write their child-pair record
__global__ void k_relayout_nodes(const rt_bvh_node* __restrict__ nodes, uint32_t nn,
    const uint32_t* __restrict__ interior_index, float4* __restrict__ out_nodes, int* __restrict__ err)
{
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nn) return;
    const rt_bvh_node nd = nodes[i];
    if ((nd.num_primitives_axis >> 16) != 0) return;
    if (interior_index[i] == RT_EMPTY_REF) return;           // not reachable from the root
    uint32_t c0 = i + 1u, c1 = nd.offset;                    // first child follows, second child at offset
    if (c0 >= nn || c1 >= nn) { atomicCAS(err, RL_OK, RL_CHILD_RANGE); return; }
    uint32_t axis = nd.num_primitives_axis & 0xFFFFu;
    if (axis > 2u) { atomicCAS(err, RL_OK, RL_AXIS); return; }
    const rt_bvh_node a = nodes[c0], b = nodes[c1];
    uint32_t r0 = (a.num_primitives_axis >> 16) ? (RT_LEAF_BIT | a.offset) : interior_index[c0];
    uint32_t r1 = (b.num_primitives_axis >> 16) ? (RT_LEAF_BIT | b.offset) : interior_index[c1];
    float4* out = out_nodes + (size_t)interior_index[i] * 4;
    out[0] = make_float4(a.bounds_min.x, a.bounds_min.y, a.bounds_min.z, a.bounds_max.x);
    out[1] = make_float4(a.bounds_max.y, a.bounds_max.z, b.bounds_min.x, b.bounds_min.y);
    out[2] = make_float4(b.bounds_min.z, b.bounds_max.x, b.bounds_max.y, b.bounds_max.z);
    out[3] = make_float4(__uint_as_float(r0), __uint_as_float(r1), __uint_as_float(axis), 0.0f);
}

// one thread per triangle: 64-byte trace record (p1, e1, e2) and 128-byte shading record
__global__ void k_relayout_triangles(const rt_triangle* __restrict__ tris, uint32_t nt, uint32_t num_materials,
    const uint8_t* __restrict__ last_in_leaf, float4* __restrict__ trt, float4* __restrict__ tsh, int* __restrict__ err)
{
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nt) return;
    const rt_triangle t = tris[i];
    const rt_float3 p1 = t.v1.position, p2 = t.v2.position, p3 = t.v3.position;
    float4* r = trt + (size_t)i * 4;
    r[0] = make_float4(p1.x, p1.y, p1.z, last_in_leaf[i] ? 1.0f : 0.0f);
    r[1] = make_float4(p2.x - p1.x, p2.y - p1.y, p2.z - p1.z, 0.0f);      // e1, trace_bvh.cl:30
    r[2] = make_float4(p3.x - p1.x, p3.y - p1.y, p3.z - p1.z, 0.0f);      // e2, trace_bvh.cl:31
    r[3] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);                           // pad to one 64-byte line
    float4* q = tsh + (size_t)i * 8;
    q[0] = make_float4(p1.x, p1.y, p1.z, t.v1.texcoord.x);
    q[1] = make_float4(p2.x, p2.y, p2.z, t.v1.texcoord.y);
    q[2] = make_float4(p3.x, p3.y, p3.z, t.v2.texcoord.x);
    q[3] = make_float4(t.v1.normal.x, t.v1.normal.y, t.v1.normal.z, t.v2.texcoord.y);
    q[4] = make_float4(t.v2.normal.x, t.v2.normal.y, t.v2.normal.z, t.v3.texcoord.x);
    q[5] = make_float4(t.v3.normal.x, t.v3.normal.y, t.v3.normal.z, t.v3.texcoord.y);
    if (t.mtl_index >= num_materials) atomicCAS(err, RL_OK, RL_MATERIAL);
    q[6] = make_float4(__uint_as_float(t.mtl_index), 0.0f, 0.0f, 0.0f);
    q[7] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
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
