// hip_pt_integrator.cpp -- binds the Integrator stage hooks to the C-ABI.
// Mapping to the reference backend (src/integrator/cl_pt_integrator.cpp):
//   ctor (buffers :188-259)            -> rt_frame_create
//   UploadGPUData :373-456             -> rt_scene_upload
//   SetCameraData :365-371             -> rt_set_camera
//   Reset :497-508                     -> rt_reset
//   GenerateRays :516-520              -> rt_generate_rays
//   IntersectRays :522-539             -> rt_intersect
//   ShadeMissedRays :582-592           -> rt_shade_miss   (no-op: fused into rt_shade)
//   ClearOutgoingRayCounter :651-657   -> rt_clear_outgoing_counter (no-op: per-bounce counters)
//   ClearShadowRayCounter :659-663     -> rt_clear_shadow_counter   (no-op)
//   ShadeSurfaceHits :594-643          -> rt_shade
//   IntersectShadowRays :564-580       -> rt_intersect_shadow (+ accumulate)
//   AccumulateDirectSamples :645-649   -> rt_accumulate_direct (no-op: fused)
//   AdvanceSampleCount :510-514        -> rt_advance_sample
//   ComputeAOVs :541-562               -> rt_compute_aovs
//   Denoise / CopyHistoryBuffers :665-675 -> rt_denoise / rt_copy_history
//   ResolveRadiance :677-684           -> rt_frame_resolve (the frame's only host sync)
#include "hip_pt_integrator.hpp"
#include <cstdio>
#include "acceleration_structure.hpp"
#include "scene.hpp"

namespace rt
{
HIPContext::HIPContext(int device_ordinal)
{
    if (rt_ctx_create(device_ordinal, &ctx_) != RT_OK)
        throw HIPException(std::string("Failed to create the HIP context: ") + rt_last_error(nullptr));
}

HIPContext::~HIPContext() { rt_ctx_destroy(ctx_); }

void HIPContext::Finish() const
{
    if (rt_finish(ctx_) != RT_OK) throw HIPException(rt_last_error(ctx_));
}

void HIPContext::LoadBlueNoiseTables(const std::string& path)
{
    const size_t n[3] = {65536, 131072, 131072};
    std::vector<unsigned char> raw(n[0] + n[1] + n[2]);
    FILE* f = fopen(path.c_str(), "rb");
    if (!f || fread(raw.data(), 1, raw.size(), f) != raw.size())
    {
        if (f) fclose(f);
        throw HIPException("Failed to load the blue-noise sampler tables " + path);
    }
    fclose(f);
    std::vector<int> t(raw.begin(), raw.end());
    if (rt_upload_blue_noise_tables(ctx_, t.data(), t.data() + n[0], t.data() + n[0] + n[1]) != RT_OK)
        throw HIPException(rt_last_error(ctx_));
    has_blue_noise_ = true;
}

std::string HIPContext::DeviceName() const
{
    char name[256] = {0};
    int cu = 0;
    size_t mem = 0;
    rt_ctx_device_info(ctx_, name, sizeof(name), &cu, &mem);
    return std::string(name) + ", " + std::to_string(cu) + " CUs, " + std::to_string(mem >> 30) + " GiB";
}

void HIPPathTraceIntegrator::Check(int rc) const
{
    if (rc != RT_OK) throw HIPException(rt_last_error(context_.Get()));
}

HIPPathTraceIntegrator::HIPPathTraceIntegrator(std::uint32_t width, std::uint32_t height,
    AccelerationStructure& acc_structure, HIPContext& context, TileDesc tile)
    : Integrator(width, height, acc_structure), context_(context)
{
    rt_frame_desc fd = {width, height, tile.rank, tile.count, tile.band_height};
    Check(rt_frame_create(context_.Get(), &fd, &frame_));
    // a constructor that throws runs no destructor: whatever follows the frame's creation gives it back itself
    try
    {
        resolved_.assign((size_t)rt_frame_local_rows(frame_) * width * 4, 0.0f);
        CreateKernels();
        // Integrate() through the hooks is this class's whole purpose: let the backend MEASURE whether a frame is better served by its stage
        // kernels or by one k_frame launch (RT_OPT_FRAME_KERNEL = 255; bit-identical either way; SetFrameKernel(0) keeps the stage kernels)
        Check(rt_set_option(frame_, RT_OPT_FRAME_KERNEL, 255u));
        // ... and trace a standing camera's next samples ahead, in batches (RT_OPT_SAMPLES_AHEAD = 1: the image after every Integrate() is the same
        // bit for bit; a launch of k samples is not its own tail the way a launch of one is; SetSamplesAhead(0) switches it off)
        Check(rt_set_option(frame_, RT_OPT_SAMPLES_AHEAD, 1u));
    }
    catch (...)
    {
        rt_frame_destroy(frame_);
        frame_ = nullptr;
        throw;
    }
    // ResolveRadiance() lands here every frame (the reference resolves into a GL image, cl_pt_integrator.cpp:677-684):
    // page-locked, the read-back runs at the PCIe rate.  Best effort -- a refusal only costs speed.  Registered LAST: nothing
    // after it can throw and leave the vector's memory freed while still page-locked.
    resolved_pinned_ = !resolved_.empty() &&
        rt_host_register(context_.Get(), resolved_.data(), resolved_.size() * sizeof(float)) == RT_OK;
}

std::vector<float> const& HIPPathTraceIntegrator::GetResolvedImage() const
{
    Check(rt_frame_present_wait(frame_));
    return resolved_;
}

HIPPathTraceIntegrator::~HIPPathTraceIntegrator()
{
    if (frame_) rt_frame_present_wait(frame_);
    if (resolved_pinned_) rt_host_unregister(context_.Get(), resolved_.data());
    rt_frame_destroy(frame_);
}

void HIPPathTraceIntegrator::UploadGPUData(Scene const& scene, AccelerationStructure const& acc_structure)
{
    auto const& nodes = acc_structure.GetNodes();
    auto const& env = scene.GetEnvImage();
    rt_scene_desc sd = {};
    sd.triangles = (const rt_triangle*)scene.GetTriangles().data();
    sd.num_triangles = (uint32_t)scene.GetTriangles().size();
    sd.nodes = (const rt_bvh_node*)nodes.data();
    sd.num_nodes = (uint32_t)nodes.size();
    sd.materials = scene.GetMaterials().data();
    sd.num_materials = (uint32_t)scene.GetMaterials().size();
    sd.textures = scene.GetTextures().data();
    sd.num_textures = (uint32_t)scene.GetTextures().size();
    sd.texture_data = scene.GetTextureData().data();
    sd.num_texture_data = (uint32_t)scene.GetTextureData().size();
    sd.lights = scene.GetLights().data();
    sd.num_lights = (uint32_t)scene.GetLights().size();
    sd.emissive_indices = scene.GetEmissiveIndices().data();
    sd.num_emissive = (uint32_t)scene.GetEmissiveIndices().size();
    // opt-in extensions (rt_scene_desc): both absent = the reference's behaviour
    sd.material_texture_indices = scene.GetMaterialTextureIndices().empty() ? nullptr : scene.GetMaterialTextureIndices().data();
    sd.flags = scene.GetEmissiveNee() ? RT_SCENE_EMISSIVE_NEE : 0u;
    sd.env_rgba = (const float*)env.data.data();
    sd.env_width = env.width;
    sd.env_height = env.height;
    Check(rt_scene_upload(context_.Get(), &sd));
}

void HIPPathTraceIntegrator::SetCameraData(Camera const& camera)
{
    prev_camera_ = camera_;
    camera_ = camera;
    Check(rt_set_camera(frame_, &camera));
}

void HIPPathTraceIntegrator::SetSamplerType(SamplerType sampler_type)
{
    if (sampler_type == sampler_type_) return;
    if (sampler_type == SamplerType::kBlueNoise && !context_.HasBlueNoiseTables())
        context_.LoadBlueNoiseTables(blue_noise_path_);
    Check(rt_set_option(frame_, RT_OPT_SAMPLER, sampler_type == SamplerType::kBlueNoise ? 1u : 0u));
    sampler_type_ = sampler_type;
    RequestReset();
}

void HIPPathTraceIntegrator::SetAOV(AOV aov)
{
    if (aov == aov_) return;
    Check(rt_set_option(frame_, RT_OPT_AOV, (uint32_t)aov));
    aov_ = aov;
    RequestReset();
}

void HIPPathTraceIntegrator::EnableDenoiser(bool enable)
{
    if (enable == enable_denoiser_) return;
    Check(rt_set_option(frame_, RT_OPT_DENOISER, enable ? 1u : 0u));
    enable_denoiser_ = enable;
    RequestReset();
}

// Kernel variants are compiled ahead of time for gfx950; "creating kernels"
// (cl_pt_integrator.cpp:261-363, a JIT build per variant) reduces to selecting them.
void HIPPathTraceIntegrator::CreateKernels() { SyncOptions(); }

void HIPPathTraceIntegrator::SyncOptions()
{
    Check(rt_set_option(frame_, RT_OPT_MAX_BOUNCES, max_bounces_));
    Check(rt_set_option(frame_, RT_OPT_WHITE_FURNACE, enable_white_furnace_ ? 1u : 0u));
}

void HIPPathTraceIntegrator::SetFrameKernel(std::uint32_t mode) { Check(rt_set_option(frame_, RT_OPT_FRAME_KERNEL, mode)); }
void HIPPathTraceIntegrator::SetSamplesAhead(std::uint32_t mode) { Check(rt_set_option(frame_, RT_OPT_SAMPLES_AHEAD, mode)); }
void HIPPathTraceIntegrator::Reset() { SyncOptions(); Check(rt_reset(frame_)); }
void HIPPathTraceIntegrator::AdvanceSampleCount() { Check(rt_advance_sample(frame_)); }
void HIPPathTraceIntegrator::GenerateRays() { SyncOptions(); Check(rt_generate_rays(frame_)); }
void HIPPathTraceIntegrator::IntersectRays(std::uint32_t bounce) { Check(rt_intersect(frame_, bounce)); }
void HIPPathTraceIntegrator::ComputeAOVs() { Check(rt_compute_aovs(frame_)); }
void HIPPathTraceIntegrator::ShadeMissedRays(std::uint32_t bounce) { Check(rt_shade_miss(frame_, bounce)); }
void HIPPathTraceIntegrator::ShadeSurfaceHits(std::uint32_t bounce) { Check(rt_shade(frame_, bounce)); }
void HIPPathTraceIntegrator::IntersectShadowRays() { Check(rt_intersect_shadow(frame_, current_bounce_)); }
void HIPPathTraceIntegrator::AccumulateDirectSamples() { Check(rt_accumulate_direct(frame_)); }
void HIPPathTraceIntegrator::ClearOutgoingRayCounter(std::uint32_t bounce) { Check(rt_clear_outgoing_counter(frame_, bounce)); }
void HIPPathTraceIntegrator::ClearShadowRayCounter() { Check(rt_clear_shadow_counter(frame_)); }
void HIPPathTraceIntegrator::Denoise() { Check(rt_denoise(frame_)); }
void HIPPathTraceIntegrator::CopyHistoryBuffers() { Check(rt_copy_history(frame_)); }

void HIPPathTraceIntegrator::ResolveRadiance()
{
    // the frame's kernels have finished when this returns (Finish(), cl_pt_integrator.cpp:682); the image travels to
    // resolved_ meanwhile and GetResolvedImage() waits for it
    if (resolve_every_frame_) Check(rt_frame_present(frame_, resolved_.data()));
}

void HIPPathTraceIntegrator::IntegrateSamples(std::uint32_t n_samples)
{
    if (request_reset_ || enable_denoiser_) { Reset(); request_reset_ = false; }
    SyncOptions();
    Check(rt_integrate(frame_, n_samples));
}

std::uint32_t HIPPathTraceIntegrator::ReserveSamples(std::uint32_t n_samples)
{
    SyncOptions();
    std::uint32_t reserved = 0;
    Check(rt_frame_reserve_samples(frame_, n_samples, &reserved));
    return reserved;
}

std::vector<float> HIPPathTraceIntegrator::ReadRadianceSum() const
{
    std::vector<float> out((size_t)rt_frame_local_rows(frame_) * width_ * 4);
    Check(rt_frame_read_radiance(frame_, out.data()));
    return out;
}

std::vector<float> const& HIPPathTraceIntegrator::ResolveNow()
{
    Check(rt_frame_resolve(frame_, resolved_.data()));
    return resolved_;
}

std::uint32_t HIPPathTraceIntegrator::GetSampleCount() const { return rt_frame_sample_count(frame_); }
std::uint32_t HIPPathTraceIntegrator::GetLocalRows() const { return rt_frame_local_rows(frame_); }
std::uint32_t HIPPathTraceIntegrator::GetGlobalRow(std::uint32_t r) const { return rt_frame_global_row(frame_, r); }

rt_stats HIPPathTraceIntegrator::GetStats() const
{
    rt_stats st;
    Check(rt_frame_get_stats(frame_, &st));
    return st;
}
} // namespace rt
