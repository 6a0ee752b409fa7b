// ref_driver.cpp -- TEST INFRASTRUCTURE.  Host driver that replays
// Integrator::Integrate() (reference src/integrator/integrator.cpp:27-59) with
// the buffer/argument wiring of CLPathTraceIntegrator
// (src/integrator/cl_pt_integrator.cpp:188-259, 373-456, 497-684) over the
// reference's UNMODIFIED OpenCL kernels compiled for x86-64 (oracle/Makefile).
// An NDRange of `work_size` items is a static-chunk parallel-for; each work
// item is one call of the kernel symbol with `ref_global_id` set.
//
// This is the strongest oracle available offline (no OpenCL device exists in
// the build container or on the GPU box) and, timed, the "reference" CPU
// baseline of bench.py.
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <vector>
#include <thread>
#include <functional>
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include "rt_types.h"

typedef float float2 __attribute__((ext_vector_type(2)));
typedef float float3 __attribute__((ext_vector_type(3)));
typedef float float4 __attribute__((ext_vector_type(4)));
typedef unsigned int uint;

struct CamCL   // by-value kernel argument; layout == Camera (shared_structures.h:173-181)
{
    float3 position, front, up;
    float fov, aspect_ratio, aperture, focus_distance;
};
static_assert(sizeof(CamCL) == 64, "Camera");
struct SceneInfoCL { uint analytic_light_count, emissive_count, environment_map_index, padding; };
struct ShimImage { int width, height; float* data; };

extern "C" {
extern thread_local size_t ref_global_id;

void ResetRadiance(uint, uint, float4*);
void ClearCounter(uint*);
void IncrementCounter(uint*);
void RayGeneration(uint, uint, CamCL, uint*, rt_ray*, uint*, uint*, float3*, float3*, float*, float3*, float2*);
void TraceBvh(rt_ray*, uint*, void* /*RTTriangle*/, rt_bvh_node*, rt_hit*);
void TraceBvhShadow(rt_ray*, uint*, void*, rt_bvh_node*, uint*);
void Miss(rt_ray*, uint*, rt_hit*, uint*, float3*, ShimImage*, float3*);
void MissFurnace(rt_ray*, uint*, rt_hit*, uint*, float3*, ShimImage*, float3*);
#define HIT_SURFACE_ARGS \
    rt_ray*, uint*, uint*, rt_hit*, rt_triangle*, rt_light*, uint*, rt_packed_material*, rt_texture*, uint*, \
    uint, uint, uint, uint*, SceneInfoCL, int*, int*, int*, float3*, rt_ray*, uint*, uint*, rt_ray*, uint*, uint*, \
    float3*, float4*
void HitSurface(HIT_SURFACE_ARGS);
void HitSurfaceFurnace(HIT_SURFACE_ARGS);
void HitSurfaceBlue(HIT_SURFACE_ARGS);
void HitSurfaceFurnaceBlue(HIT_SURFACE_ARGS);
void AccumulateDirectSamples(uint*, uint*, uint*, float3*, float4*);
void ResolveRadiance(uint, uint, uint, float4*, float3*, float*, float3*, float2*, uint*, ShimImage*);
void ResolveRadianceDenoiser(uint, uint, uint, float4*, float3*, float*, float3*, float2*, uint*, ShimImage*);
void GenerateAOV(rt_ray*, uint*, uint*, rt_hit*, rt_triangle*, rt_packed_material*, rt_texture*, uint*, uint, uint, CamCL, CamCL,
    float3*, float*, float3*, float2*);
void TemporalAccumulation(uint, uint, float4*, float4*, float*, float*, float2*);
}

namespace
{
struct RTTri { rt_float3 p1, p2, p3; };   // RTTriangle, shared_structures.h:143-153
class Pool;

struct RefIntegrator
{
    uint width = 0, height = 0;
    int furnace = 0;
    int threads = 1;
    uint max_bounces = 3;       // integrator.hpp:91
    bool request_reset = false;
    CamCL camera = {};
    CamCL prev_camera = {};        // Integrator::prev_camera_ (integrator.hpp:89)
    CamCL aov_prev_camera = {};    // the kPrevCamera argument bound by the last SetCameraData
    int denoiser = 0;
    int blue_noise = 0;
    std::vector<int> bn_sobol, bn_scramble, bn_rank;   // rank padded with 256 zeros (sampling.h:50 overrun)
    uint aov = 0;
    SceneInfoCL scene_info = {};
    std::vector<rt_float4> prev_radiance;
    std::vector<float> prev_depth;

    // per-pixel buffers (cl_pt_integrator.cpp:199-249)
    std::vector<rt_float4> radiance;
    std::vector<rt_ray> rays[2];
    std::vector<uint> pixel_indices[2];
    // every counter is its own cl::Buffer in the reference (cl_pt_integrator.cpp:206-219):
    // keep them on separate cache lines so that the atomic appends of HitSurface do not
    // false-share with the read-only counters
    alignas(64) uint ray_counter_a = 0;
    alignas(64) uint ray_counter_b = 0;
    uint& rc(uint i) { return i ? ray_counter_b : ray_counter_a; }
    std::vector<rt_ray> shadow_rays;
    std::vector<uint> shadow_pixel_indices;
    alignas(64) uint shadow_ray_counter = 0;
    std::vector<rt_hit> hits;
    std::vector<uint> shadow_hits;
    std::vector<rt_float3> throughputs;
    alignas(64) uint sample_counter = 0;
    alignas(64) char pad_after_counters[64] = {0};
    std::vector<rt_float4> direct_light_samples;
    std::vector<rt_float3> diffuse_albedo, normal;
    std::vector<float> depth;
    std::vector<rt_float2> velocity;
    std::vector<rt_float4> resolved;

    // scene (cl_pt_integrator.cpp:373-456)
    std::vector<rt_triangle> triangles;
    std::vector<RTTri> rt_triangles;
    std::vector<rt_bvh_node> nodes;
    std::vector<rt_packed_material> materials;
    std::vector<rt_texture> textures;
    std::vector<uint> texture_data;
    std::vector<rt_light> lights;
    std::vector<uint> emissive;
    std::vector<float> env;
    ShimImage env_image = {0, 0, nullptr};

    // statistics (ray counters sampled per bounce)
    uint64_t total_closest = 0, total_shadow = 0;
    uint last_active[64] = {0}, last_shadow[64] = {0};
    std::shared_ptr<Pool> pool;
};

// single work-item launches (ExecuteKernel(kernel, 1), cl_pt_integrator.cpp:503,513,656,662)
void Clear(uint* counter) { ref_global_id = 0; ClearCounter(counter); }
void Increment(uint* counter) { ref_global_id = 0; IncrementCounter(counter); }

// Persistent worker pool: an NDRange is split into 256-item chunks handed out
// through an atomic counter (dynamic schedule -- path lengths vary a lot).
class Pool
{
public:
    explicit Pool(int n) : n_(n > 1 ? n : 1)
    {
        for (int t = 1; t < n_; ++t) workers_.emplace_back([this]() { Loop(); });
    }
    ~Pool()
    {
        {
            std::unique_lock<std::mutex> lk(m_);
            stop_ = true;
            ++epoch_;
        }
        cv_.notify_all();
        for (auto& w : workers_) w.join();
    }
    void Run(size_t work_size, const std::function<void()>& body)
    {
        if (n_ == 1 || work_size < 4096)
        {
            for (size_t i = 0; i < work_size; ++i) { ref_global_id = i; body(); }
            return;
        }
        {
            std::unique_lock<std::mutex> lk(m_);
            body_ = &body;
            work_ = work_size;
            next_.store(0);
            pending_ = n_ - 1;
            ++epoch_;
        }
        cv_.notify_all();
        Work();
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this]() { return pending_ == 0; });
    }

private:
    void Work()
    {
        const size_t chunk = 256;
        for (;;)
        {
            size_t b = next_.fetch_add(chunk);
            if (b >= work_) break;
            size_t e = b + chunk < work_ ? b + chunk : work_;
            for (size_t i = b; i < e; ++i) { ref_global_id = i; (*body_)(); }
        }
    }
    void Loop()
    {
        uint64_t seen = 0;
        for (;;)
        {
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&]() { return epoch_ != seen; });
                seen = epoch_;
                if (stop_) return;
            }
            Work();
            std::unique_lock<std::mutex> lk(m_);
            if (--pending_ == 0) done_.notify_one();
        }
    }
    int n_;
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void()>* body_ = nullptr;
    size_t work_ = 0;
    std::atomic<size_t> next_{0};
    int pending_ = 0;
    uint64_t epoch_ = 0;
    bool stop_ = false;
};

template <class F>
void NDRange(RefIntegrator& r, size_t work_size, F&& body)
{
    if (!r.pool) r.pool = std::make_shared<Pool>(r.threads);
    std::function<void()> fn = body;
    r.pool->Run(work_size, fn);
}

void Reset(RefIntegrator& r)                       // cl_pt_integrator.cpp:497-508
{
    if (!r.denoiser) Clear(&r.sample_counter);
    NDRange(r, r.width * r.height, [&]() { ResetRadiance(r.width, r.height, (float4*)r.radiance.data()); });
}

void GenerateRays(RefIntegrator& r)                // :516-520
{
    NDRange(r, r.width * r.height, [&]()
    {
        RayGeneration(r.width, r.height, r.camera, &r.sample_counter, r.rays[0].data(), &r.rc(0),
            r.pixel_indices[0].data(), (float3*)r.throughputs.data(), (float3*)r.diffuse_albedo.data(),
            r.depth.data(), (float3*)r.normal.data(), (float2*)r.velocity.data());
    });
}

void IntersectRays(RefIntegrator& r, uint bounce)  // :522-539
{
    uint in = bounce & 1;
    NDRange(r, r.width * r.height, [&]()
    {
        TraceBvh(r.rays[in].data(), &r.rc(in), r.rt_triangles.data(), r.nodes.data(), r.hits.data());
    });
}

void ShadeMissedRays(RefIntegrator& r, uint bounce) // :582-592
{
    uint in = bounce & 1;
    NDRange(r, r.width * r.height, [&]()
    {
        if (r.furnace)
            MissFurnace(r.rays[in].data(), &r.rc(in), r.hits.data(), r.pixel_indices[in].data(),
                (float3*)r.throughputs.data(), &r.env_image, (float3*)r.radiance.data());
        else
            Miss(r.rays[in].data(), &r.rc(in), r.hits.data(), r.pixel_indices[in].data(),
                (float3*)r.throughputs.data(), &r.env_image, (float3*)r.radiance.data());
    });
}

void ShadeSurfaceHits(RefIntegrator& r, uint bounce) // :594-643
{
    uint in = bounce & 1, out = (bounce + 1) & 1;
    auto fn = r.blue_noise ? (r.furnace ? HitSurfaceFurnaceBlue : HitSurfaceBlue)
                           : (r.furnace ? HitSurfaceFurnace : HitSurface);
    NDRange(r, r.width * r.height, [&]()
    {
        fn(r.rays[in].data(), &r.rc(in), r.pixel_indices[in].data(), r.hits.data(),
            r.triangles.data(), r.lights.data(), r.emissive.data(), r.materials.data(),
            r.textures.data(), r.texture_data.data(), bounce, r.width, r.height, &r.sample_counter,
            r.scene_info, r.bn_sobol.data(), r.bn_scramble.data(), r.bn_rank.data(), (float3*)r.throughputs.data(),
            r.rays[out].data(), &r.rc(out), r.pixel_indices[out].data(),
            r.shadow_rays.data(), &r.shadow_ray_counter, r.shadow_pixel_indices.data(),
            (float3*)r.direct_light_samples.data(), (float4*)r.radiance.data());
    });
}

void IntersectShadowRays(RefIntegrator& r)         // :564-580
{
    NDRange(r, r.width * r.height, [&]()
    {
        TraceBvhShadow(r.shadow_rays.data(), &r.shadow_ray_counter, r.rt_triangles.data(), r.nodes.data(),
            r.shadow_hits.data());
    });
}

void ComputeAOVs(RefIntegrator& r)                 // :541-562
{
    NDRange(r, r.width * r.height, [&]()
    {
        GenerateAOV(r.rays[0].data(), &r.rc(0), r.pixel_indices[0].data(), r.hits.data(), r.triangles.data(),
            r.materials.data(), r.textures.data(), r.texture_data.data(), r.width, r.height, r.camera, r.aov_prev_camera,
            (float3*)r.diffuse_albedo.data(), r.depth.data(), (float3*)r.normal.data(), (float2*)r.velocity.data());
    });
}

void Denoise(RefIntegrator& r)                     // :665-668
{
    NDRange(r, r.width * r.height, [&]()
    {
        TemporalAccumulation(r.width, r.height, (float4*)r.radiance.data(), (float4*)r.prev_radiance.data(),
            r.depth.data(), r.prev_depth.data(), (float2*)r.velocity.data());
    });
}

void CopyHistory(RefIntegrator& r)                 // :670-675
{
    r.prev_radiance = r.radiance;
    r.prev_depth = r.depth;
}

void AccumulateDirect(RefIntegrator& r)            // :645-649
{
    NDRange(r, r.width * r.height, [&]()
    {
        AccumulateDirectSamples(r.shadow_hits.data(), &r.shadow_ray_counter, r.shadow_pixel_indices.data(),
            (float3*)r.direct_light_samples.data(), (float4*)r.radiance.data());
    });
}
} // namespace

extern "C" {

void* ref_create(uint32_t width, uint32_t height, int white_furnace, int threads)
{
    auto* r = new RefIntegrator;
    r->width = width; r->height = height; r->furnace = white_furnace;
    r->threads = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    size_t n = (size_t)width * height;
    r->radiance.resize(n);
    for (int i = 0; i < 2; ++i) { r->rays[i].resize(n); r->pixel_indices[i].resize(n); }
    r->shadow_rays.resize(n); r->shadow_pixel_indices.resize(n);
    r->hits.resize(n); r->shadow_hits.resize(n); r->throughputs.resize(n);
    r->direct_light_samples.resize(n);
    r->diffuse_albedo.resize(n); r->normal.resize(n); r->depth.resize(n); r->velocity.resize(n);
    r->resolved.resize(n);
    r->prev_radiance.resize(n); r->prev_depth.resize(n);
    Reset(*r);                                     // ctor ends with Reset(), :258
    return r;
}

void ref_destroy(void* h) { delete (RefIntegrator*)h; }

void ref_upload(void* h, const rt_triangle* tris, uint32_t ntris, const rt_bvh_node* nodes, uint32_t nnodes,
    const rt_packed_material* mats, uint32_t nmats, const rt_texture* tex, uint32_t ntex,
    const uint32_t* texdata, uint32_t ntexdata, const rt_light* lights, uint32_t nlights,
    const uint32_t* emissive, uint32_t nemissive, const float* env_rgba, uint32_t env_w, uint32_t env_h)
{
    auto& r = *(RefIntegrator*)h;
    r.triangles.assign(tris, tris + ntris);
    r.rt_triangles.resize(ntris);                  // :392-402
    for (uint32_t i = 0; i < ntris; ++i)
    {
        r.rt_triangles[i].p1 = tris[i].v1.position;
        r.rt_triangles[i].p2 = tris[i].v2.position;
        r.rt_triangles[i].p3 = tris[i].v3.position;
    }
    r.nodes.assign(nodes, nodes + nnodes);
    r.materials.assign(mats, mats + nmats);
    r.textures.assign(tex, tex + ntex);
    r.texture_data.assign(texdata, texdata + ntexdata);
    r.lights.assign(lights, lights + nlights);
    r.emissive.assign(emissive, emissive + nemissive);
    r.env.assign(env_rgba, env_rgba + (size_t)env_w * env_h * 4);
    r.env_image = ShimImage{(int)env_w, (int)env_h, r.env.data()};
    r.scene_info = SceneInfoCL{nlights, nemissive, 0, 0};   // scene.cpp:338,358
}

void ref_set_camera(void* h, const rt_camera* cam)
{
    // CLPathTraceIntegrator::SetCameraData, cl_pt_integrator.cpp:365-371
    auto& r = *(RefIntegrator*)h;
    memcpy(&r.camera, cam, sizeof(CamCL));
    r.aov_prev_camera = r.prev_camera;
    r.prev_camera = r.camera;
}

void ref_enable_denoiser(void* h, int enable)      // :485-495
{
    auto& r = *(RefIntegrator*)h;
    if ((enable != 0) == (r.denoiser != 0)) return;
    r.denoiser = enable != 0;
    r.request_reset = true;
}

void ref_set_blue_noise_tables(void* h, const int* sobol, const int* scrambling, const int* ranking)   // :222-235
{
    auto& r = *(RefIntegrator*)h;
    r.bn_sobol.assign(sobol, sobol + 65536);
    r.bn_scramble.assign(scrambling, scrambling + 131072);
    r.bn_rank.assign(ranking, ranking + 131072);
    r.bn_rank.resize(131072 + 256, 0);
}

void ref_set_sampler(void* h, int blue_noise)      // SetSamplerType, :458-468
{
    auto& r = *(RefIntegrator*)h;
    if ((blue_noise != 0) == (r.blue_noise != 0)) return;
    r.blue_noise = blue_noise != 0;
    r.request_reset = true;
}

void ref_set_aov(void* h, uint32_t aov)            // :470-483
{
    auto& r = *(RefIntegrator*)h;
    if (aov == r.aov) return;
    r.aov = aov;
    r.request_reset = true;
}

void ref_set_max_bounces(void* h, uint32_t b)      // integrator.cpp:61-65
{
    auto& r = *(RefIntegrator*)h;
    r.max_bounces = b; r.request_reset = true;
}
void ref_request_reset(void* h) { ((RefIntegrator*)h)->request_reset = true; }

// stage entry points (the 15 protected virtuals, integrator.hpp:65-79)
void ref_stage_reset(void* h) { Reset(*(RefIntegrator*)h); }
void ref_stage_generate_rays(void* h) { GenerateRays(*(RefIntegrator*)h); }
void ref_stage_intersect(void* h, uint32_t b) { IntersectRays(*(RefIntegrator*)h, b); }
void ref_stage_shade_miss(void* h, uint32_t b) { ShadeMissedRays(*(RefIntegrator*)h, b); }
void ref_stage_clear_counters(void* h, uint32_t b)
{
    auto& r = *(RefIntegrator*)h;
    Clear(&r.rc((b + 1) & 1));     // :651-657
    Clear(&r.shadow_ray_counter);           // :659-663
}
void ref_stage_shade_hits(void* h, uint32_t b) { ShadeSurfaceHits(*(RefIntegrator*)h, b); }
void ref_stage_intersect_shadow(void* h) { IntersectShadowRays(*(RefIntegrator*)h); }
void ref_stage_accumulate(void* h) { AccumulateDirect(*(RefIntegrator*)h); }
void ref_stage_advance(void* h) { Increment(&((RefIntegrator*)h)->sample_counter); }

// Integrator::Integrate(), integrator.cpp:27-59 (denoiser off, AOV stage skipped)
void ref_integrate(void* h)
{
    auto& r = *(RefIntegrator*)h;
    if (r.request_reset || r.denoiser) { Reset(r); r.request_reset = false; }
    GenerateRays(r);
    for (uint bounce = 0; bounce <= r.max_bounces; ++bounce)
    {
        IntersectRays(r, bounce);
        if (bounce == 0) ComputeAOVs(r);
        ShadeMissedRays(r, bounce);
        Clear(&r.rc((bounce + 1) & 1));
        Clear(&r.shadow_ray_counter);
        ShadeSurfaceHits(r, bounce);
        IntersectShadowRays(r);
        AccumulateDirect(r);
        uint active = r.rc(bounce & 1);
        r.total_closest += active;
        r.total_shadow += r.shadow_ray_counter;
        if (bounce < 64) { r.last_active[bounce] = active; r.last_shadow[bounce] = r.shadow_ray_counter; }
    }
    Increment(&r.sample_counter);
    if (r.denoiser)
    {
        Denoise(r);
        CopyHistory(r);
    }
}

// ResolveRadiance (resolve_radiance.cl:31-86), headless image
const float* ref_resolve(void* h)
{
    auto& r = *(RefIntegrator*)h;
    ShimImage out{(int)r.width, (int)r.height, (float*)r.resolved.data()};
    NDRange(r, r.width * r.height, [&]()
    {
        (r.denoiser ? ResolveRadianceDenoiser : ResolveRadiance)(r.width, r.height, r.aov, (float4*)r.radiance.data(),
            (float3*)r.diffuse_albedo.data(), r.depth.data(), (float3*)r.normal.data(), (float2*)r.velocity.data(),
            &r.sample_counter, &out);
    });
    return (const float*)r.resolved.data();
}

const float* ref_radiance(void* h) { return (const float*)((RefIntegrator*)h)->radiance.data(); }
uint32_t ref_sample_count(void* h) { return ((RefIntegrator*)h)->sample_counter; }
void ref_ray_totals(void* h, uint64_t* closest, uint64_t* shadow)
{
    *closest = ((RefIntegrator*)h)->total_closest; *shadow = ((RefIntegrator*)h)->total_shadow;
}
void ref_last_counts(void* h, uint32_t* active, uint32_t* shadow, uint32_t n)
{
    auto& r = *(RefIntegrator*)h;
    for (uint32_t i = 0; i < n && i < 64; ++i) { active[i] = r.last_active[i]; shadow[i] = r.last_shadow[i]; }
}

// raw buffer access for per-stage parity tests
void* ref_buffer(void* h, const char* name)
{
    auto& r = *(RefIntegrator*)h;
    if (!strcmp(name, "rays0")) return r.rays[0].data();
    if (!strcmp(name, "rays1")) return r.rays[1].data();
    if (!strcmp(name, "pixel_indices0")) return r.pixel_indices[0].data();
    if (!strcmp(name, "pixel_indices1")) return r.pixel_indices[1].data();
    if (!strcmp(name, "ray_counter0")) return &r.rc(0);
    if (!strcmp(name, "ray_counter1")) return &r.rc(1);
    if (!strcmp(name, "shadow_rays")) return r.shadow_rays.data();
    if (!strcmp(name, "shadow_pixel_indices")) return r.shadow_pixel_indices.data();
    if (!strcmp(name, "shadow_ray_counter")) return &r.shadow_ray_counter;
    if (!strcmp(name, "hits")) return r.hits.data();
    if (!strcmp(name, "shadow_hits")) return r.shadow_hits.data();
    if (!strcmp(name, "throughputs")) return r.throughputs.data();
    if (!strcmp(name, "direct_light_samples")) return r.direct_light_samples.data();
    if (!strcmp(name, "radiance")) return r.radiance.data();
    if (!strcmp(name, "sample_counter")) return &r.sample_counter;
    return nullptr;
}

} // extern "C"
