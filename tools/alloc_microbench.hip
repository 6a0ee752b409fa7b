// What mapping the per-path buffers costs, and when: hipMalloc / hipFree by size in a fresh process, again right after a large free and again after a pause (does the
// driver wipe released pages before they can be handed out again?), as twelve blocks, through the stream-ordered allocator, through the virtual-memory calls in 1 GiB
// chunks, and from a second thread while the first launches kernels (does an allocation stall the launches beside it?).
// Round 6: bench.py's cold_job showed 0.2 - 4.3 s for buffers of 36 - 150 GB in a process that had just freed its batch buffers; rt_render in a fresh process did not.
// Build: hipcc --offload-arch=gfx950 -O2 tools/alloc_microbench.hip -o tools/bin/alloc_mb -lpthread        Run: tools/bin/alloc_mb [GiB of the large block = 100]
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(char* p, size_t n, size_t stride) { size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * stride; if (i < n) p[i] = 1; }
__global__ void spin(unsigned* out, unsigned n) { unsigned a = threadIdx.x; for (unsigned i = 0; i < n; ++i) a = a * 1664525u + 1013904223u; if (a == 7u) out[0] = a; }
static double timed_malloc(void** p, size_t bytes, hipError_t* e) { double t0 = now(); *e = hipMalloc(p, bytes); return now() - t0; }
static void touch_all(void* p, size_t bytes) { hipLaunchKernelGGL(touch, dim3((unsigned)((bytes / (2u << 20) + 255) / 256)), dim3(256), 0, 0, (char*)p, bytes, (size_t)2 << 20); hipDeviceSynchronize(); }
int main(int argc, char** argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 100.0;
    const size_t G = (size_t)1 << 30, total = (size_t)(gib * (double)G);
    hipFree(0);
    hipError_t e;
    // 1. by size, in a process that has freed nothing yet (every block kept until the end of the sweep: no released page can be handed out again)
    {
        std::vector<void*> keep;
        for (size_t g : { (size_t)1, (size_t)4, (size_t)16, (size_t)32, (size_t)64 })
        {
            void* p = nullptr; double t = timed_malloc(&p, g * G, &e); double t1 = now(); if (e == hipSuccess) touch_all(p, g * G);
            printf("fresh process, nothing freed yet: hipMalloc of %3zu GiB %.3f s (%.1f GiB/s; %s), first touch of every 2 MiB %.3f s\n", g, t, (double)g / t, hipGetErrorString(e), now() - t1);
            keep.push_back(p);
        }
        double t0 = now(); for (void* p : keep) hipFree(p); printf("hipFree of the five blocks (117 GiB): %.3f s\n", now() - t0);
    }
    // 2. right after that free, and after a pause
    for (int rep = 0; rep < 3; ++rep)
    {
        void* p = nullptr; double t = timed_malloc(&p, total, &e); double t1 = now();
        printf("%s: one block of %.0f GiB: hipMalloc %.3f s (%s)", rep == 0 ? "right after the free" : rep == 1 ? "right after the next free" : "5 s after the free", gib, t, hipGetErrorString(e));
        if (e == hipSuccess) { touch_all(p, total); double t2 = now(); hipFree(p); printf(", first touch %.3f s, hipFree %.3f s", t2 - t1, now() - t2); }
        printf("\n");
        if (rep == 1) std::this_thread::sleep_for(std::chrono::seconds(5));
    }
    {
        double t0 = now(); std::vector<void*> v(12, nullptr); for (auto& q : v) hipMalloc(&q, total / 12); double t1 = now();
        for (auto q : v) hipFree(q); double t2 = now();
        printf("twelve blocks of %.1f GiB right after a free: hipMalloc %.3f s, hipFree %.3f s\n", gib / 12.0, t1 - t0, t2 - t1);
        void* p = nullptr; double t = timed_malloc(&p, 8 * G, &e); printf("8 GiB right after that free: hipMalloc %.3f s\n", t); if (e == hipSuccess) hipFree(p);
        std::this_thread::sleep_for(std::chrono::seconds(5));
        t = timed_malloc(&p, 8 * G, &e); printf("8 GiB 5 s later: hipMalloc %.3f s\n", t); if (e == hipSuccess) hipFree(p);
    }
    // 3. the stream-ordered allocator with a pool that keeps what it is given back
    {
        hipStream_t st; hipStreamCreate(&st);
        hipMemPool_t pool; hipDeviceGetDefaultMemPool(&pool, 0);
        unsigned long long thr = ~0ull; hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
        for (int rep = 0; rep < 2; ++rep)
        {
            double t0 = now(); void* p = nullptr; e = hipMallocAsync(&p, total, st); hipStreamSynchronize(st); double t1 = now();
            if (e == hipSuccess) { hipFreeAsync(p, st); hipStreamSynchronize(st); }
            double t2 = now();
            printf("rep %d: hipMallocAsync of one block of %.0f GiB %.3f s (%s), hipFreeAsync %.3f s\n", rep, gib, t1 - t0, hipGetErrorString(e), t2 - t1);
        }
        hipMemPoolTrimTo(pool, 0); hipStreamDestroy(st);
    }
    // 4. virtual-memory calls: one reservation, physical chunks of 1 GiB mapped one after the other (a buffer that grows in place)
    {
        hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        size_t gran = 0; e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
        printf("virtual memory: granularity %zu (%s)\n", gran, hipGetErrorString(e));
        void* va = nullptr; double t0 = now(); e = hipMemAddressReserve(&va, 32 * G, 0, nullptr, 0); printf("hipMemAddressReserve of 32 GiB: %.4f s (%s)\n", now() - t0, hipGetErrorString(e));
        if (e == hipSuccess)
        {
            std::vector<hipMemGenericAllocationHandle_t> hs; double tc = 0, tm = 0, ta = 0;
            for (int i = 0; i < 32; ++i)
            {
                hipMemGenericAllocationHandle_t h; double a = now(); if (hipMemCreate(&h, G, &prop, 0) != hipSuccess) break; double b = now();
                if (hipMemMap((char*)va + (size_t)i * G, G, 0, h, 0) != hipSuccess) break; double c = now();
                hipMemAccessDesc ad = {}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
                if (hipMemSetAccess((char*)va + (size_t)i * G, G, &ad, 1) != hipSuccess) break; double d = now();
                tc += b - a; tm += c - b; ta += d - c; hs.push_back(h);
            }
            printf("%zu chunks of 1 GiB: hipMemCreate %.3f s, hipMemMap %.3f s, hipMemSetAccess %.3f s in all\n", hs.size(), tc, tm, ta);
            if (!hs.empty()) { double a = now(); touch_all(va, hs.size() * G); printf("first touch of the mapped range %.3f s\n", now() - a); }
            double a = now(); for (size_t i = 0; i < hs.size(); ++i) { hipMemUnmap((char*)va + i * G, G); hipMemRelease(hs[i]); } hipMemAddressFree(va, 32 * G);
            printf("unmap + release + free of the reservation: %.3f s\n", now() - a);
        }
    }
    // 5. a second thread allocates while this one launches 100 us kernels back to back: the longest gap between two completions, with and without the allocation beside them
    {
        unsigned* out = nullptr; hipMalloc(&out, 4); hipStream_t st; hipStreamCreate(&st);
        for (int with = 0; with < 2; ++with)
        {
            std::atomic<bool> go{ false }, done{ false }; double t_alloc = 0;
            std::thread th([&] { while (!go.load()) {} if (with) { void* p = nullptr; hipError_t ee; t_alloc = timed_malloc(&p, 48 * G, &ee); double f0 = now(); if (ee == hipSuccess) hipFree(p); t_alloc += 0 * (now() - f0); } done.store(true); });
            double worst = 0, sum = 0; int n = 0; go.store(true); const double t_begin = now();
            while (!done.load() || now() - t_begin < 1.0)
            {
                double a = now(); hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, out, 20000u); hipStreamSynchronize(st); double d = now() - a;
                if (d > worst) worst = d; sum += d; ++n;
            }
            th.join();
            printf("%s: %d launch + sync rounds, mean %.3f ms, longest %.3f ms%s", with ? "48 GiB hipMalloc + hipFree on a second thread" : "nothing beside", n, 1e3 * sum / n, 1e3 * worst, with ? "" : "\n");
            if (with) printf(" (the hipMalloc itself: %.3f s)\n", t_alloc);
        }
        hipFree(out); hipStreamDestroy(st);
    }
    return 0;
}
