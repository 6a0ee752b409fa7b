// What does a frame's image on its way to the host cost the kernels that run meanwhile?  (Round 5: in the per-frame timeline k_raygen takes 0.41 ms
// instead of 0.02 whenever the previous frame's resolved image -- 33 MB, hipMemcpyAsync to page-locked memory on a low-priority stream -- is still
// travelling: profiles/r05_final_per_frame_gantt.log.)  A 33 MB device-to-host copy into (a) hipHostMalloc'ed memory, (b) malloc'ed + hipHostRegister'ed
// memory (what HIPPathTraceIntegrator does with its resolved_ vector), alone; a raygen-like writer kernel (100 MB of stores) alone; and the writer
// launched while each copy is in flight.  Run under rocprofv3 --kernel-trace to see which copies are shader blits (__amd_rocclr_copyBuffer).
// build: hipcc --offload-arch=gfx950 -O3 tools/d2h_copy_probe.hip -o tools/bin/d2h_copy_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void writer(float4* a, float4* b, float4* c, unsigned n)
{
    unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    a[i] = make_float4(i, 1, 2, 3); b[i] = make_float4(4, i, 6, 7); c[i] = make_float4(1, 1, 1, 0);
}
__global__ void reader(const float4* a, float* out, unsigned n)        // a trace-like kernel: reads, few writes
{
    unsigned i = blockIdx.x * 256u + threadIdx.x;
    float s = 0;
    for (unsigned k = 0; k < 64; ++k) { float4 v = a[(i * 97u + k * 8191u) % n]; s += v.x + v.w; }
    if (s == 12345.678f) out[i % 64] = s;
}

int main()
{
    const size_t image = (size_t)1920 * 1080 * 16;
    const unsigned n = 1920 * 1080;
    float4 *dimg, *a, *b, *c; float* dout;
    CK(hipMalloc(&dimg, image)); CK(hipMalloc(&a, image)); CK(hipMalloc(&b, image)); CK(hipMalloc(&c, image)); CK(hipMalloc(&dout, 256));
    void* hA; CK(hipHostMalloc(&hA, image, hipHostMallocDefault));
    void* hB = aligned_alloc(4096, image); memset(hB, 0, image); CK(hipHostRegister(hB, image, hipHostRegisterDefault));
    int lo = 0, hi = 0; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, lo));
    hipEvent_t e0, e1, c0, c1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&c0)); CK(hipEventCreate(&c1));
    auto ms = [&](hipEvent_t x, hipEvent_t y) { float t = 0; hipEventElapsedTime(&t, x, y); return t; };
    for (int rep = 0; rep < 3; ++rep)
    {
        hipLaunchKernelGGL(writer, dim3((n + 255) / 256), dim3(256), 0, s1, a, b, c, n);
        hipLaunchKernelGGL(reader, dim3((n + 255) / 256), dim3(256), 0, s1, a, dout, n);
    }
    CK(hipDeviceSynchronize());
    for (int which = 0; which < 2; ++which)
    {
        void* h = which ? hB : hA;
        const char* name = which ? "hipHostRegister'ed" : "hipHostMalloc'ed  ";
        float copy_alone = 0, w_alone = 0, r_alone = 0, w_with = 0, r_with = 0, copy_with_w = 0, copy_with_r = 0;
        for (int rep = 0; rep < 5; ++rep)
        {
            CK(hipEventRecord(c0, s2)); CK(hipMemcpyAsync(h, dimg, image, hipMemcpyDeviceToHost, s2)); CK(hipEventRecord(c1, s2)); CK(hipDeviceSynchronize());
            copy_alone = ms(c0, c1);
            CK(hipEventRecord(e0, s1)); hipLaunchKernelGGL(writer, dim3((n + 255) / 256), dim3(256), 0, s1, a, b, c, n); CK(hipEventRecord(e1, s1)); CK(hipDeviceSynchronize());
            w_alone = ms(e0, e1);
            CK(hipEventRecord(e0, s1)); hipLaunchKernelGGL(reader, dim3((n + 255) / 256), dim3(256), 0, s1, a, dout, n); CK(hipEventRecord(e1, s1)); CK(hipDeviceSynchronize());
            r_alone = ms(e0, e1);
            // the writer while the copy is in flight
            CK(hipEventRecord(c0, s2)); CK(hipMemcpyAsync(h, dimg, image, hipMemcpyDeviceToHost, s2)); CK(hipEventRecord(c1, s2));
            CK(hipEventRecord(e0, s1)); hipLaunchKernelGGL(writer, dim3((n + 255) / 256), dim3(256), 0, s1, a, b, c, n); CK(hipEventRecord(e1, s1)); CK(hipDeviceSynchronize());
            w_with = ms(e0, e1); copy_with_w = ms(c0, c1);
            CK(hipEventRecord(c0, s2)); CK(hipMemcpyAsync(h, dimg, image, hipMemcpyDeviceToHost, s2)); CK(hipEventRecord(c1, s2));
            CK(hipEventRecord(e0, s1)); hipLaunchKernelGGL(reader, dim3((n + 255) / 256), dim3(256), 0, s1, a, dout, n); CK(hipEventRecord(e1, s1)); CK(hipDeviceSynchronize());
            r_with = ms(e0, e1); copy_with_r = ms(c0, c1);
        }
        // ... and as rt_frame_present issues it: a kernel on s1 writes the image, s2 waits for that kernel's event, then copies; the writer follows on s1
        float w_dep = 0, copy_dep = 0;
        hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        for (int rep = 0; rep < 5; ++rep)
        {
            hipLaunchKernelGGL(writer, dim3((n + 255) / 256), dim3(256), 0, s1, dimg, b, c, n);
            CK(hipEventRecord(ev, s1)); CK(hipStreamWaitEvent(s2, ev, 0));
            CK(hipEventRecord(c0, s2)); CK(hipMemcpyAsync(h, dimg, image, hipMemcpyDeviceToHost, s2)); CK(hipEventRecord(c1, s2));
            CK(hipStreamSynchronize(s1));
            CK(hipEventRecord(e0, s1)); hipLaunchKernelGGL(writer, dim3((n + 255) / 256), dim3(256), 0, s1, a, b, c, n); CK(hipEventRecord(e1, s1)); CK(hipDeviceSynchronize());
            w_dep = ms(e0, e1); copy_dep = ms(c0, c1);
        }
        printf("%s: the copy waiting for a kernel's event on another stream (rt_frame_present's sequence): writer meanwhile %.3f ms, copy %.3f ms\n", name, w_dep, copy_dep);
        printf("%s: copy alone %.3f ms (%.1f GB/s); writer alone %.3f ms, while the copy is in flight %.3f ms (copy then %.3f); reader alone %.3f ms, with the copy %.3f ms (copy then %.3f)\n",
            name, copy_alone, image / copy_alone / 1e6, w_alone, w_with, copy_with_w, r_alone, r_with, copy_with_r);
    }
    return 0;
}
