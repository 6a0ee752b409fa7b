// Micro-benchmark: dependent random 64-byte record fetches, one record per lane per
// iteration (the access pattern of k_trace).
//   mode 0: each lane issues 4 x global_load_dwordx4 on its own record (64 distinct
//           addresses per instruction)
//   mode 1: cooperative -- 4 adjacent lanes fetch one lane's record with ONE
//           global_load_lds_dwordx4 (16 contiguous-64B requests per instruction),
//           4 instructions cover the wave, data lands in LDS and every lane reads
//           its own record back with 4 x ds_read_b128
//   mode 2: as mode 1, but (a) the record indices travel through LDS as one ds_write_b32 +
//           4 broadcast ds_read_b32 instead of 8 bpermutes, and (b) the piece each loader lane
//           fetches is rotated by (record >> 2) so that the readback at a 64-byte lane stride
//           is bank-conflict free
// Build: hipcc --offload-arch=gfx950 -O3 tools/fetch_microbench.hip -o /tmp/fetch_mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(64) void k_fetch(const float4* __restrict__ recs, unsigned n_recs, unsigned iters,
    unsigned coherence, float4* __restrict__ out)
{
    __shared__ float4 stage[4][64];
    __shared__ unsigned idxs[64];
    const unsigned lane = threadIdx.x;
    unsigned idx = (blockIdx.x * 64u + lane) * 2654435761u;
    if (coherence) idx = (blockIdx.x * 2654435761u) + lane / coherence;   // groups of lanes share records
    idx %= n_recs;
    float4 acc = make_float4(0, 0, 0, 0);
    for (unsigned it = 0; it < iters; ++it)
    {
        float4 q0, q1, q2, q3;
        const float4* base = recs + (size_t)idx * 4;
        if (MODE == 0)
        {
            q0 = base[0]; q1 = base[1]; q2 = base[2]; q3 = base[3];
        }
        else if (MODE == 2)
        {
            idxs[lane] = idx;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned piece = ((lane & 3u) - (lane >> 4)) & 3u;
            for (int k = 0; k < 4; ++k)
            {
                unsigned us = idxs[16 * k + (lane >> 2)];
                const float4* p = recs + (size_t)us * 4 + piece;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                    (__attribute__((address_space(3))) void*)&stage[k][0], 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const float4* mine = &stage[0][0] + lane * 4;
            const unsigned rot = lane >> 2;
            q0 = mine[(0 + rot) & 3]; q1 = mine[(1 + rot) & 3]; q2 = mine[(2 + rot) & 3]; q3 = mine[(3 + rot) & 3];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        else
        {
            unsigned long long A = (unsigned long long)base;
            for (int k = 0; k < 4; ++k)
            {
                int r = 16 * k + (lane >> 2);
                unsigned lo = __shfl((unsigned)A, r, 64), hi = __shfl((unsigned)(A >> 32), r, 64);
                const char* p = (const char*)(((unsigned long long)hi << 32) | lo) + (lane & 3) * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                    (__attribute__((address_space(3))) void*)&stage[k][0], 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const float4* mine = &stage[lane >> 4][(lane & 15) * 4];
            q0 = mine[0]; q1 = mine[1]; q2 = mine[2]; q3 = mine[3];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        acc.x += q0.x + q1.y; acc.y += q2.z; acc.z += q3.w; acc.w += q1.x;
        // dependent next index (like a BVH child reference)
        idx = (__float_as_uint(q3.x) + it * 7u + (coherence ? 0u : lane)) % n_recs;
    }
    out[blockIdx.x * 64u + lane] = acc;
}

int main(int argc, char** argv)
{
    unsigned n_recs = argc > 1 ? atoi(argv[1]) : 1700000, iters = 200, blocks = 256 * 16;
    std::vector<float4> h((size_t)n_recs * 4);
    unsigned s = 12345;
    for (size_t i = 0; i < h.size(); ++i)
    {
        s = s * 1664525u + 1013904223u;
        unsigned v = (s >> 5) % n_recs;              // high LCG bits: the low ones cycle quickly
        float f; memcpy(&f, &v, 4);
        h[i] = make_float4(f, (float)(i & 255), 1.0f, 2.0f);
    }
    float4 *d, *out;
    hipMalloc(&d, h.size() * 16); hipMalloc(&out, (size_t)blocks * 64 * 16);
    hipMemcpy(d, h.data(), h.size() * 16, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    std::vector<float4> r0((size_t)blocks * 64), r1((size_t)blocks * 64);
    for (unsigned coh : {0u, 4u, 16u})
        for (int mode = 0; mode < 3; ++mode)
        {
            for (int rep = 0; rep < 2; ++rep)
            {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(k_fetch<0>, dim3(blocks), dim3(64), 0, 0, d, n_recs, iters, coh, out);
                else if (mode == 2) hipLaunchKernelGGL(k_fetch<2>, dim3(blocks), dim3(64), 0, 0, d, n_recs, iters, coh, out);
                else hipLaunchKernelGGL(k_fetch<1>, dim3(blocks), dim3(64), 0, 0, d, n_recs, iters, coh, out);
                hipEventRecord(b); hipEventSynchronize(b);
            }
            float ms; hipEventElapsedTime(&ms, a, b);
            hipMemcpy(mode ? r1.data() : r0.data(), out, r0.size() * 16, hipMemcpyDeviceToHost);
            double fetches = (double)blocks * 64 * iters;
            printf("coherence %2u mode %d: %.3f ms, %.2f G records/s, %.2f TB/s of 64-byte records\n", coh, mode, ms,
                fetches / ms / 1e6, fetches * 64 / ms / 1e9);
            if (mode >= 1) printf("   results identical: %s\n", memcmp(r0.data(), r1.data(), r0.size() * 16) == 0 ? "yes" : "NO");
        }
    return 0;
}
