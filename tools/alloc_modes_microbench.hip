// One way of obtaining a large device buffer per process (so that nothing is cached from another way): what the call costs, what the first full write costs, and how fast
// the memory then is -- a streaming read of all of it and a random gather of 64-byte records over all of it (the page size behind the mapping shows in the gather).
// Build: hipcc --offload-arch=gfx950 -O2 tools/alloc_modes_microbench.hip -o tools/bin/alloc_modes        Run: tools/bin/alloc_modes <malloc|async|vmm|vmm2m|seq-malloc|seq-async> [GiB = 100]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_stream(const uint4* p, size_t n, unsigned* out)
{
    uint4 a = make_uint4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; a.x ^= v.x; a.y ^= v.y; a.z ^= v.z; a.w ^= v.w; }
    if ((a.x ^ a.y ^ a.z ^ a.w) == 0x12345u) out[0] = 1;
}
__global__ void k_gather(const uint4* p, size_t n_records, unsigned per_thread, unsigned* out)
{
    unsigned long long s = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345ull; uint4 a = make_uint4(0, 0, 0, 0);
    for (unsigned k = 0; k < per_thread; ++k)
    {
        s = s * 6364136223846793005ull + 1442695040888963407ull; size_t r = (size_t)((s >> 20) % n_records);
        uint4 v0 = p[r * 4], v1 = p[r * 4 + 1], v2 = p[r * 4 + 2], v3 = p[r * 4 + 3];
        a.x ^= v0.x ^ v1.y ^ v2.z ^ v3.w;
    }
    if (a.x == 0x12345u) out[0] = 1;
}
int main(int argc, char** argv)
{
    const char* mode = argc > 1 ? argv[1] : "malloc"; const double gib = argc > 2 ? atof(argv[2]) : 100.0;
    const size_t G = (size_t)1 << 30, total = (size_t)(gib * (double)G) / G * G;
    CK(hipFree(0)); hipStream_t st; CK(hipStreamCreate(&st)); unsigned* out = nullptr; CK(hipMalloc(&out, 4));
    if (!strncmp(mode, "seq-", 4))
    {
        // the bench's sequence in one process: a kernel first, a large block written and freed, a block of half the size, the large block again
        const bool pool_mode = !strcmp(mode, "seq-async");
        if (pool_mode) { hipMemPool_t pool; CK(hipDeviceGetDefaultMemPool(&pool, 0)); unsigned long long thr = ~0ull; CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr)); }
        hipLaunchKernelGGL(k_stream, dim3(1), dim3(64), 0, st, (const uint4*)out, 0, out); CK(hipStreamSynchronize(st));
        const size_t sizes[6] = { total, total / 2, total, total / 4, total / 2 + total / 4, total };
        for (int i = 0; i < 6; ++i)
        {
            void* q = nullptr; double a = now();
            if (pool_mode) { CK(hipMallocAsync(&q, sizes[i], st)); CK(hipStreamSynchronize(st)); } else CK(hipMalloc(&q, sizes[i]));
            double b = now(); CK(hipMemsetAsync(q, 0x11 * (i + 1), sizes[i], st)); CK(hipStreamSynchronize(st)); double c = now();
            if (pool_mode) { CK(hipFreeAsync(q, st)); CK(hipStreamSynchronize(st)); } else CK(hipFree(q));
            printf("%-9s step %d: %5.1f GiB obtain %.3f s, full write %.3f s, release %.3f s\n", mode, i, (double)sizes[i] / (double)G, b - a, c - b, now() - c);
        }
        return 0;
    }
    void* p = nullptr; double t0 = now();
    if (!strcmp(mode, "malloc")) { CK(hipMalloc(&p, total)); }
    else if (!strcmp(mode, "async"))
    {
        hipMemPool_t pool; CK(hipDeviceGetDefaultMemPool(&pool, 0)); unsigned long long thr = ~0ull; CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
        CK(hipMallocAsync(&p, total, st)); CK(hipStreamSynchronize(st));
    }
    else
    {
        hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        const size_t chunk = !strcmp(mode, "vmm2m") ? ((size_t)2 << 20) : G;
        CK(hipMemAddressReserve(&p, total, (size_t)2 << 20, nullptr, 0));
        for (size_t off = 0; off < total; off += chunk) { hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, chunk, &prop, 0)); CK(hipMemMap((char*)p + off, chunk, 0, h, 0)); CK(hipMemRelease(h)); }
        hipMemAccessDesc ad = {}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite; CK(hipMemSetAccess(p, total, &ad, 1));
    }
    double t1 = now();
    CK(hipMemsetAsync(p, 0x5a, total, st)); CK(hipStreamSynchronize(st)); double t2 = now();
    CK(hipMemsetAsync(p, 0x3c, total, st)); CK(hipStreamSynchronize(st)); double t3 = now();
    printf("%-7s %5.0f GiB: obtain %.3f s, first full write %.3f s, second full write %.3f s (%.0f GB/s)\n", mode, gib, t1 - t0, t2 - t1, t3 - t2, (double)total / (t3 - t2) / 1e9);
    for (int rep = 0; rep < 2; ++rep)
    {
        double a = now(); hipLaunchKernelGGL(k_stream, dim3(256 * 16), dim3(256), 0, st, (const uint4*)p, total / 16, out); CK(hipStreamSynchronize(st)); double b = now();
        const unsigned per = 64; const size_t threads = (size_t)256 * 256 * 64;
        hipLaunchKernelGGL(k_gather, dim3((unsigned)(threads / 256)), dim3(256), 0, st, (const uint4*)p, total / 64, per, out); CK(hipStreamSynchronize(st)); double c = now();
        printf("  rep %d: streaming read %.0f GB/s; random 64-byte gathers over all of it %.2f G records/s (%.0f GB/s)\n", rep, (double)total / (b - a) / 1e9, (double)threads * per / (c - b) / 1e9,
            (double)threads * per * 64.0 / (c - b) / 1e9);
    }
    double f0 = now();
    if (!strcmp(mode, "malloc")) CK(hipFree(p)); else if (!strcmp(mode, "async")) { CK(hipFreeAsync(p, st)); CK(hipStreamSynchronize(st)); } else { CK(hipMemUnmap(p, total)); CK(hipMemAddressFree(p, total)); }
    printf("  release %.3f s\n", now() - f0);
    return 0;
}
