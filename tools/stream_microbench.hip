// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against KNOWN byte counts, per access pattern of the hot path
// (the 0.99 factor of tools/make_counters_json.py was calibrated on random 64-byte records -- k_trace_w4's
// pattern -- and then applied to k_shade, whose traffic is coalesced 16 B / lane streams; /opt/skills/guides/MI355X_MICROARCH.md says
// FETCH_SIZE reports HALF of such a read on this rocprofv3).
// Every kernel moves exactly `bytes` (printed), far past the 256 MiB Infinity Cache, once:
//   k_read16    coalesced 16 B / lane reads           (k_shade's d4 / thr / hits streams, k_trace's queue reads)
//   k_write16   coalesced 16 B / lane writes          (k_shade's outgoing-ray and shadow-ray streams, k_raygen)
//   k_write12   coalesced 12 B / lane writes, as 3 dwords at a 12-byte lane stride (log_store: the radiance log's entries)
//   k_read12    ... read back the same way            (k_flush)
//   k_rand64    one random 64-byte record per lane, 4 x dwordx4 (k_trace_w4's nodes / triangles; the old calibration)
//   k_gather128 one random 128-byte record per lane, 8 x dwordx4 (k_shade's shading-triangle gather)
// usage: stream_mb [GiB per kernel, default 2]      then:  rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum -- stream_mb
//                                                           rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum -- stream_mb
// tools/stream_calibration.py turns the two passes into profiles/r06_fetch_size_calibration.json.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k_read16(const float4* __restrict__ src, size_t n, float4* __restrict__ sink)
{
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u)
    {
        const float4 v = src[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x == 12345.678f) sink[threadIdx.x] = acc;           // never true: keeps the loads
}

__global__ __launch_bounds__(256) void k_write16(float4* __restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u)
        dst[i] = make_float4((float)i, 1.f, 2.f, 3.f);
}

__global__ __launch_bounds__(256) void k_write12(float* __restrict__ dst, size_t n_entries)
{
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n_entries; i += (size_t)gridDim.x * 256u)
    {
        dst[3 * i] = (float)i; dst[3 * i + 1] = 1.f; dst[3 * i + 2] = 2.f;
    }
}

__global__ __launch_bounds__(256) void k_read12(const float* __restrict__ src, size_t n_entries, float* __restrict__ sink)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n_entries; i += (size_t)gridDim.x * 256u)
        acc += src[3 * i] + src[3 * i + 1] + src[3 * i + 2];
    if (acc == 12345.678f) sink[threadIdx.x] = acc;
}

template <int QUADS>
__global__ __launch_bounds__(256) void k_rand(const float4* __restrict__ recs, unsigned n_recs, unsigned per_lane, float4* __restrict__ sink)
{
    unsigned idx = ((blockIdx.x * 256u + threadIdx.x) * 2654435761u) % n_recs;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (unsigned it = 0; it < per_lane; ++it)
    {
        const float4* p = recs + (size_t)idx * QUADS;
#pragma unroll
        for (int k = 0; k < QUADS; ++k) { const float4 v = p[k]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        idx = (idx * 1664525u + 1013904223u + (unsigned)acc.w) % n_recs;   // (acc.w is 0: the dependence is only formal)
    }
    if (acc.x == 12345.678f) sink[threadIdx.x] = acc;
}

int main(int argc, char** argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 2.0;
    const size_t bytes = (size_t)(gib * 1024.0 * 1024.0 * 1024.0) / 768u * 768u;      // a multiple of 16, 12, 64 and 128
    char* buf = nullptr; float4* sink = nullptr;
    if (hipMalloc((void**)&buf, bytes) != hipSuccess || hipMalloc((void**)&sink, 4096) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    const unsigned blocks = 256u * 32u;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto timed = [&](const char* name, double moved, auto&& launch)
    {
        hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0.f; hipEventElapsedTime(&ms, a, b);
        printf("%-12s bytes %.0f  ms %.3f  TB/s %.3f\n", name, moved, ms, moved / ms / 1e9);
    };
    for (int rep = 0; rep < 2; ++rep)     // (two launches of each: the counter tool averages them)
    {
        timed("k_read16", (double)bytes, [&]() { hipLaunchKernelGGL(k_read16, dim3(blocks), dim3(256), 0, 0, (const float4*)buf, bytes / 16, sink); });
        timed("k_write16", (double)bytes, [&]() { hipLaunchKernelGGL(k_write16, dim3(blocks), dim3(256), 0, 0, (float4*)buf, bytes / 16); });
        timed("k_write12", (double)bytes, [&]() { hipLaunchKernelGGL(k_write12, dim3(blocks), dim3(256), 0, 0, (float*)buf, bytes / 12); });
        timed("k_read12", (double)bytes, [&]() { hipLaunchKernelGGL(k_read12, dim3(blocks), dim3(256), 0, 0, (const float*)buf, bytes / 12, (float*)sink); });
        // random records: every lane fetches `per_lane` records of a table as large as the buffer (hit rates ~ 0)
        const unsigned per_lane = 64;
        const double moved64 = (double)blocks * 256.0 * per_lane * 64.0, moved128 = (double)blocks * 256.0 * per_lane * 128.0;
        timed("k_rand64", moved64, [&]() { hipLaunchKernelGGL(k_rand<4>, dim3(blocks), dim3(256), 0, 0, (const float4*)buf, (unsigned)(bytes / 64), per_lane, sink); });
        timed("k_gather128", moved128, [&]() { hipLaunchKernelGGL(k_rand<8>, dim3(blocks), dim3(256), 0, 0, (const float4*)buf, (unsigned)(bytes / 128), per_lane, sink); });
    }
    hipFree(buf); hipFree(sink);
    return 0;
}
