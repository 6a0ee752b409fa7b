// visit_microbench.hip -- the LATENCY ceiling of the wide-tree walk (VERDICT r02 "next round" 2): the dependent chain of ONE
// wide-node visit of k_trace_w4's loop C, on its own and at the kernel's residency.
//
//   fetch the 64-byte record of the current node (4 x global_load_dwordx4 through an SGPR base + 32-bit offset)
//   -> w4_test_slots (the kernel's own code, raytracing_amd/csrc/trace_kernels.h: dequantise, 4 slab tests, visit order)
//   -> push the later passing slots to the per-lane LDS stack, pop when nothing passes
//   -> the NEXT node depends on all of it
//
// Every lane runs such a chain over a table of random wide nodes; where the next record lies is drawn so that a stated
// fraction of the fetches hit L1 (a 16 KiB hot set PER COMPUTE UNIT: the wave reads which CU it runs on -- HW_ID / XCC_ID --
// so the hot set stays one CU's however many waves share it; round 3 tied it to the block index, and at the kernel's own
// residency of 26 waves per CU that put 26 hot sets behind one L1: the curve collapsed exactly where it mattered), L2 (the
// 32 hot sets + a 2 MiB set per XCD) or neither (1 GiB table) -- the hit mix of the production kernel from its PMC counters
// (profiles/r0X_trace_counters.json: l1_hit_rate, l2_hit_rate).
// Reported per configuration (waves per CU, hit mix): ns per visit of a lane's chain, and visits per second of the machine.
// bench.py turns that into  ceiling = resident lanes x lane utilisation / visit latency / steps per ray.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Iraytracing_amd/csrc tools/visit_microbench.hip -o tools/bin/visit_mb
// usage: tools/bin/visit_mb [l1_hit l2_hit] [steps] [1 = also the quad-cooperative fetch variant]   (defaults 0.93 0.85 4096 0)   prints one JSON object
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
#include "trace_kernels.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Rec { float ox, oy, oz; uint32_t meta; uint32_t lo[3]; uint32_t hi[3]; uint32_t ref[4]; uint32_t order; uint32_t pad; };
static_assert(sizeof(Rec) == 64, "wide node record");

// one wave per block, 12-entry LDS stack like the production instance (6 KiB per wave -> 26 waves per CU)
// COOP: quad-cooperative fetch -- load instruction j of a lane reads the quarter (lane & 3) of the record quad member j
// wants, so the four lanes of a quad read ONE 64-byte record per instruction (one L1 look-up instead of four); the quarters
// are handed to their owners through a padded LDS staging buffer (4 x ds_write_b128, 4 x ds_read_b128, 4 KiB per wave).
template <bool SHADOW, bool COOP = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_visit_chain(const float4* __restrict__ nodes, uint32_t n_hot,
    uint32_t n_l2, uint32_t n_all, uint32_t thr_l1, uint32_t thr_l2, uint32_t steps, float* __restrict__ out)
{
    __shared__ uint2 stack[12][64];                  // 6144 bytes exactly, like the production instance: 26 waves per CU fit (round 3's build declared a
                                                     // 16-byte dummy beside it for the non-cooperative instances -- one LDS granule more, 24 per CU, and the
                                                     // 26-waves point of its sweep ran in two rounds: THAT was the collapse at the kernel's own residency)
    extern __shared__ float4 stage_dyn[];            // COOP only: 4 x 65 float4, passed as dynamic shared memory
    float4 (*stage)[65] = reinterpret_cast<float4 (*)[65]>(stage_dyn);
    const uint32_t lane = threadIdx.x;
    const char* const node_base = reinterpret_cast<const char*>(nodes);
    // a ray per lane: origin inside the unit cube the nodes live in, direction from a hash
    uint32_t h = (blockIdx.x * 64u + lane) * 2654435761u + 12345u;
    auto rnd = [&]() { h ^= h << 13; h ^= h >> 17; h ^= h << 5; return h; };
    const f3 org = F3((rnd() & 0xFFFF) * (1.0f / 65536.0f), (rnd() & 0xFFFF) * (1.0f / 65536.0f), (rnd() & 0xFFFF) * (1.0f / 65536.0f));
    f3 dir = F3((rnd() & 0xFFFF) * (2.0f / 65536.0f) - 1.0f, (rnd() & 0xFFFF) * (2.0f / 65536.0f) - 1.0f, (rnd() & 0xFFFF) * (2.0f / 65536.0f) - 1.0f);
    dir.x = dir.x == 0.0f ? 0.5f : dir.x; dir.y = dir.y == 0.0f ? 0.5f : dir.y; dir.z = dir.z == 0.0f ? 0.5f : dir.z;
    const f3 inv = F3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
    const uint32_t sign_bits = (inv.x < 0.0f ? 1u : 0u) | (inv.y < 0.0f ? 2u : 0u) | (inv.z < 0.0f ? 4u : 0u);
    const uint32_t octant4 = 4u * sign_bits;
    const float t_min = 0.0f, INF = __builtin_inff();
    float t_max = 1.0e4f;
    int sp = 0;
    // the hot set of the compute unit this wave runs on (gfx9 HW_ID: cu_id 11:8, sh_id 12, se_id 15:13; XCC_ID 3:0)
    uint32_t hw_id, xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    const uint32_t cu_slot = ((xcc_id & 0xFu) << 8) | ((hw_id >> 8) & 0xFFu);           // < 4096
    const uint32_t hot_base = cu_slot * n_hot;
    // the L2-resident set of this XCD (each XCD has an L2 of its own): after the hot sets of all CUs
    const uint32_t l2_base = 4096u * n_hot + (xcc_id & 0xFu) * n_l2;
    uint32_t ref = hot_base + (lane % n_hot);
    float acc = 0.0f;
    for (uint32_t s = 0; s < steps; ++s)
    {
        float4 q0, q1, q2, q3;
        if (COOP)
        {
            const uint32_t m = lane & 3u, qb = lane & ~3u;
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j)
            {
                const uint32_t ref_j = (uint32_t)__shfl((int)ref, (int)(qb + j), 64);
                stage[m][qb + j] = *reinterpret_cast<const float4*>(node_base + (size_t)(ref_j << 6) + (m << 4));
            }
            __syncthreads();                                                 // one-wave block: orders the LDS writes before the reads
            q0 = stage[0][lane]; q1 = stage[1][lane]; q2 = stage[2][lane]; q3 = stage[3][lane];
            __syncthreads();
        }
        else
        {
            const float4* np = reinterpret_cast<const float4*>(node_base + (size_t)(ref << 6));
            q0 = np[0]; q1 = np[1]; q2 = np[2]; q3 = np[3];
        }
        uint32_t r[4];
        float e[4];
        w4_test_slots<SHADOW>(q0, q1, q2, q3, org, inv, sign_bits, octant4, t_min, t_max, r, e);
        // the kernel's direct-visit step: later passing slots to the stack, the first one next, pop when none passes
        const bool v0 = e[0] < INF, v1 = e[1] < INF, v2 = e[2] < INF, v3 = e[3] < INF;
        uint32_t next;
        if (v3 && (v0 || v1 || v2)) { stack[sp % 12][lane] = make_uint2(r[3], __float_as_uint(e[3])); ++sp; }
        if (v2 && (v0 || v1)) { stack[sp % 12][lane] = make_uint2(r[2], __float_as_uint(e[2])); ++sp; }
        if (v1 && v0) { stack[sp % 12][lane] = make_uint2(r[1], __float_as_uint(e[1])); ++sp; }
        if (v0) next = r[0];
        else if (v1) next = r[1];
        else if (v2) next = r[2];
        else if (v3) next = r[3];
        else
        {
            next = ref * 2246822519u + s;                                    // stack empty: "the next ray"
            while (sp > 0)
            {
                --sp;
                const uint2 en = stack[sp % 12][lane];
                if (t_max >= __uint_as_float(en.y)) { next = en.x; break; }
            }
        }
        acc += e[0] < INF ? e[0] : 0.0f;
        if (sp > 9) sp = 3;                                                  // keep the synthetic stack shallow
        // where the next record lies: L1-hot set / L2-resident set / anywhere, by the hit mix asked for
        const uint32_t pick = (next ^ (next >> 15)) * 2654435761u;
        const uint32_t where = pick >> 8;                                    // 24 bits
        ref = where < thr_l1 ? hot_base + (pick % n_hot) : (where < thr_l2 ? l2_base + (pick % n_l2) : (pick % n_all));
    }
    out[blockIdx.x * 64u + lane] = acc + (float)sp;
}

// TWO independent chains per lane ("two node fetches in flight per lane"): the same visit, twice per pass,
// the two fetches issued together -- does the memory-level parallelism buy visits per second at the kernel's residency, or do the
// L1 / texture-address path and the vector ALU (both near their ceilings in the production kernel) take it back?
template <bool SHADOW>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_visit_chain2(const float4* __restrict__ nodes, uint32_t n_hot,
    uint32_t n_l2, uint32_t n_all, uint32_t thr_l1, uint32_t thr_l2, uint32_t steps, float* __restrict__ out)
{
    __shared__ uint2 stack[12][64];
    const uint32_t lane = threadIdx.x;
    const char* const node_base = reinterpret_cast<const char*>(nodes);
    uint32_t h = (blockIdx.x * 64u + lane) * 2654435761u + 12345u;
    auto rnd = [&]() { h ^= h << 13; h ^= h >> 17; h ^= h << 5; return h; };
    const f3 org = F3((rnd() & 0xFFFF) * (1.0f / 65536.0f), (rnd() & 0xFFFF) * (1.0f / 65536.0f), (rnd() & 0xFFFF) * (1.0f / 65536.0f));
    f3 dir = F3((rnd() & 0xFFFF) * (2.0f / 65536.0f) - 1.0f, (rnd() & 0xFFFF) * (2.0f / 65536.0f) - 1.0f, (rnd() & 0xFFFF) * (2.0f / 65536.0f) - 1.0f);
    dir.x = dir.x == 0.0f ? 0.5f : dir.x; dir.y = dir.y == 0.0f ? 0.5f : dir.y; dir.z = dir.z == 0.0f ? 0.5f : dir.z;
    const f3 inv = F3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
    const uint32_t sign_bits = (inv.x < 0.0f ? 1u : 0u) | (inv.y < 0.0f ? 2u : 0u) | (inv.z < 0.0f ? 4u : 0u);
    const uint32_t octant4 = 4u * sign_bits;
    const float t_min = 0.0f, INF = __builtin_inff();
    const float t_max = 1.0e4f;
    int sp = 0;
    uint32_t hw_id, xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    const uint32_t hot_base = (((xcc_id & 0xFu) << 8) | ((hw_id >> 8) & 0xFFu)) * n_hot;
    const uint32_t l2_base = 4096u * n_hot + (xcc_id & 0xFu) * n_l2;
    uint32_t ref[2] = {hot_base + (lane % n_hot), hot_base + ((lane * 7u + 3u) % n_hot)};
    float acc = 0.0f;
    for (uint32_t s = 0; s < steps; s += 2u)
    {
        const float4* np0 = reinterpret_cast<const float4*>(node_base + (size_t)(ref[0] << 6));
        const float4* np1 = reinterpret_cast<const float4*>(node_base + (size_t)(ref[1] << 6));
        const float4 a0 = np0[0], a1 = np0[1], a2 = np0[2], a3 = np0[3];
        const float4 b0 = np1[0], b1 = np1[1], b2 = np1[2], b3 = np1[3];
#pragma unroll
        for (int c = 0; c < 2; ++c)
        {
            uint32_t r[4];
            float e[4];
            if (c == 0) w4_test_slots<SHADOW>(a0, a1, a2, a3, org, inv, sign_bits, octant4, t_min, t_max, r, e);
            else w4_test_slots<SHADOW>(b0, b1, b2, b3, org, inv, sign_bits, octant4, t_min, t_max, r, e);
            const bool v0 = e[0] < INF, v1 = e[1] < INF, v2 = e[2] < INF, v3 = e[3] < INF;
            uint32_t next;
            if (v3 && (v0 || v1 || v2)) { stack[sp % 12][lane] = make_uint2(r[3], __float_as_uint(e[3])); ++sp; }
            if (v2 && (v0 || v1)) { stack[sp % 12][lane] = make_uint2(r[2], __float_as_uint(e[2])); ++sp; }
            if (v1 && v0) { stack[sp % 12][lane] = make_uint2(r[1], __float_as_uint(e[1])); ++sp; }
            if (v0) next = r[0];
            else if (v1) next = r[1];
            else if (v2) next = r[2];
            else if (v3) next = r[3];
            else
            {
                next = ref[c] * 2246822519u + s;
                while (sp > 0)
                {
                    --sp;
                    const uint2 en = stack[sp % 12][lane];
                    if (t_max >= __uint_as_float(en.y)) { next = en.x; break; }
                }
            }
            acc += e[0] < INF ? e[0] : 0.0f;
            if (sp > 9) sp = 3;
            const uint32_t pick = (next ^ (next >> 15)) * 2654435761u;
            const uint32_t where = pick >> 8;
            ref[c] = where < thr_l1 ? hot_base + (pick % n_hot) : (where < thr_l2 ? l2_base + (pick % n_l2) : (pick % n_all));
        }
    }
    out[blockIdx.x * 64u + lane] = acc + (float)sp;
}

int main(int argc, char** argv)
{
    const double l1_hit = argc > 2 ? atof(argv[1]) : 0.93, l2_hit = argc > 2 ? atof(argv[2]) : 0.85;
    const uint32_t steps = argc > 3 ? (uint32_t)atoi(argv[3]) : 4096u;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const uint32_t cus = (uint32_t)prop.multiProcessorCount;
    const uint32_t n_hot = 256u;                    // 16 KiB per neighbourhood
    const uint32_t n_l2 = 32768u;                   // 2 MiB per XCD (half of its L2)
    const uint32_t n_all = 1u << 24;                // 1 GiB
    std::vector<Rec> host((size_t)n_all);
    std::mt19937 gen(7);
    for (size_t i = 0; i < host.size(); ++i)
    {
        Rec& r = host[i];
        r.ox = 0.0f; r.oy = 0.0f; r.oz = 0.0f;
        r.meta = 119u | 119u << 8 | 119u << 16;     // cell 2^-8: the node spans the unit cube
        for (int a = 0; a < 3; ++a)
        {
            r.lo[a] = 0; r.hi[a] = 0;
            for (int k = 0; k < 4; ++k)
            {
                uint32_t lo = gen() % 200u, hi = lo + 8u + gen() % 48u;     // a slot covers ~1/6 of each axis: ~1.2 of 4 pass, as measured
                if (gen() % 3u == 0u) { lo = 0; hi = 255; }                  // ... with some that span the axis
                r.lo[a] |= lo << (8 * k); r.hi[a] |= (hi > 255u ? 255u : hi) << (8 * k);
            }
        }
        for (int k = 0; k < 4; ++k) r.ref[k] = gen() & (n_all - 1u);
        r.order = gen();
        r.pad = 0;
    }
    float4* d_nodes = nullptr;
    float* d_out = nullptr;
    CHECK(hipMalloc((void**)&d_nodes, host.size() * sizeof(Rec)));
    CHECK(hipMemcpy(d_nodes, host.data(), host.size() * sizeof(Rec), hipMemcpyHostToDevice));
    CHECK(hipMalloc((void**)&d_out, (size_t)cus * 32 * 64 * sizeof(float)));
    const uint32_t thr_l1 = (uint32_t)(l1_hit * 16777216.0), thr_l2 = thr_l1 + (uint32_t)((1.0 - l1_hit) * l2_hit * 16777216.0);
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    // how many of these one-wave blocks (6144 bytes of LDS, like the production instance) the runtime says fit on a CU
    int resident = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, k_visit_chain<false>, 64, 0));
    printf("{\"device\": \"%s\", \"compute_units\": %u, \"l1_hit\": %.4f, \"l2_hit\": %.4f, \"steps\": %u, \"resident_blocks_per_cu\": %d, \"runs\": [", prop.gcnArchName, cus, l1_hit, l2_hit, steps, resident);
    bool first = true;
    const bool coop_too = argc > 4 && atoi(argv[4]) != 0;
    for (int shadow = 0; shadow < (coop_too ? 5 : 4); ++shadow)
        for (uint32_t wpc : {1u, 4u, 8u, 12u, 16u, 20u, 24u, 25u, 26u})
        {
            if (shadow == 4 && wpc > 16u) continue;   // the staging buffer: 10 KiB of LDS per wave, 15 waves per CU
            const uint32_t blocks = cus * wpc;
            float ms = 0.0f;
            for (int rep = 0; rep < 2; ++rep)       // the first run warms the caches
            {
                CHECK(hipEventRecord(a));
                if (shadow == 2) hipLaunchKernelGGL(k_visit_chain2<false>, dim3(blocks), dim3(64), 0, 0, d_nodes, n_hot, n_l2, n_all, thr_l1, thr_l2, steps, d_out);
                else if (shadow == 3) hipLaunchKernelGGL(k_visit_chain2<true>, dim3(blocks), dim3(64), 0, 0, d_nodes, n_hot, n_l2, n_all, thr_l1, thr_l2, steps, d_out);
                else if (shadow == 4) hipLaunchKernelGGL((k_visit_chain<false, true>), dim3(blocks), dim3(64), 4 * 65 * sizeof(float4), 0, d_nodes, n_hot, n_l2, n_all, thr_l1, thr_l2, steps, d_out);
                else if (shadow) hipLaunchKernelGGL(k_visit_chain<true>, dim3(blocks), dim3(64), 0, 0, d_nodes, n_hot, n_l2, n_all, thr_l1, thr_l2, steps, d_out);
                else hipLaunchKernelGGL(k_visit_chain<false>, dim3(blocks), dim3(64), 0, 0, d_nodes, n_hot, n_l2, n_all, thr_l1, thr_l2, steps, d_out);
                CHECK(hipEventRecord(b));
                CHECK(hipEventSynchronize(b));
                CHECK(hipEventElapsedTime(&ms, a, b));
            }
            const double ns_per_visit = (double)ms * 1e6 / steps;
            const double visits_per_s = (double)blocks * 64.0 * steps / ((double)ms * 1e-3);
            printf("%s{\"kernel\": \"%s\", \"waves_per_cu\": %u, \"ms\": %.4f, \"ns_per_visit\": %.2f, \"gvisits_per_s\": %.3f}", first ? "" : ", ",
                shadow == 4 ? "closest, quad-cooperative fetch" : (shadow == 3 ? "shadow, two chains per lane" : (shadow == 2 ? "closest, two chains per lane" : (shadow ? "shadow" : "closest"))), wpc, ms, ns_per_visit, visits_per_s * 1e-9);
            first = false;
        }
    printf("]}\n");
    return 0;
}
