// hip_pt_integrator.hpp -- the MI355X backend of Integrator: the class that
// takes the place of CLPathTraceIntegrator (reference:
// src/integrator/cl_pt_integrator.{hpp,cpp}) and talks to the device only
// through the C-ABI of include/rt_hip.h.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>
#include "integrator.hpp"
#include "rt_hip.h"

namespace rt
{
// What CLException is to the OpenCL backend (src/utils/cl_exception.hpp:109-123).
class HIPException : public std::runtime_error
{
public:
    explicit HIPException(const std::string& what) : std::runtime_error(what) {}
};

// RAII owner of an rt_ctx (CLContext's role, src/gpu_wrappers/cl_context.hpp:37-65).
class HIPContext
{
public:
    explicit HIPContext(int device_ordinal = 0);
    ~HIPContext();
    HIPContext(const HIPContext&) = delete;
    HIPContext& operator=(const HIPContext&) = delete;
    rt_ctx* Get() const { return ctx_; }
    void Finish() const;
    std::string DeviceName() const;
    // Loads the blue-noise sampler tables (the data the reference compiles in through
    // src/utils/blue_noise_sampler.hpp) from a packed asset file and uploads them.
    void LoadBlueNoiseTables(const std::string& path);
    bool HasBlueNoiseTables() const { return has_blue_noise_; }

private:
    rt_ctx* ctx_ = nullptr;
    bool has_blue_noise_ = false;
};

struct TileDesc   // which interleaved row bands of the image this integrator renders
{
    std::uint32_t rank = 0, count = 1, band_height = 8;
};

class HIPPathTraceIntegrator : public Integrator
{
public:
    HIPPathTraceIntegrator(std::uint32_t width, std::uint32_t height, AccelerationStructure& acc_structure,
        HIPContext& context, TileDesc tile = TileDesc());
    ~HIPPathTraceIntegrator() override;

    void UploadGPUData(Scene const& scene, AccelerationStructure const& acc_structure) override;
    void SetCameraData(Camera const& camera) override;
    void SetSamplerType(SamplerType sampler_type) override;
    void SetAOV(AOV aov) override;
    void EnableDenoiser(bool enable) override;

    // Fast path: n x Integrate() without leaving the native side between stages.
    void IntegrateSamples(std::uint32_t n_samples);
    // Sizes the per-path device buffers for IntegrateSamples(n_samples) ahead of time; returns the
    // number of samples the device will trace together.
    std::uint32_t ReserveSamples(std::uint32_t n_samples);
    // Headless outputs (the reference writes a GL-shared image in ResolveRadiance).
    std::vector<float> const& GetResolvedImage() const;   // local_rows x width x RGBA; waits for the image of the last frame to arrive
    std::vector<float> ReadRadianceSum() const;
    std::vector<float> const& ResolveNow();      // runs the resolve stage and returns the image
    std::uint32_t GetSampleCount() const;
    std::uint32_t GetLocalRows() const;
    std::uint32_t GetGlobalRow(std::uint32_t local_row) const;
    rt_stats GetStats() const;
    void SetResolveEveryFrame(bool enable) { resolve_every_frame_ = enable; }
    // RT_OPT_FRAME_KERNEL: 255 (this class's default) = measured choice between the stage kernels and one k_frame launch per Integrate(); 0 / 1 = forced
    void SetFrameKernel(std::uint32_t mode);
    // RT_OPT_SAMPLES_AHEAD: 1 (this class's default) = while the camera stands still the backend traces the next samples ahead in batches and an
    // Integrate() whose sample is there only replays it (same radiance after every call); 0 = every Integrate() traces its own sample; k = batch size
    void SetSamplesAhead(std::uint32_t mode);
    // packed uint8 tables, see tools/make_blue_noise_asset.py (default: relative to the CWD like the env map)
    void SetBlueNoiseTablePath(std::string path) { blue_noise_path_ = std::move(path); }
    rt_frame* GetFrame() const { return frame_; }

protected:
    void CreateKernels() override;
    void Reset() override;
    void AdvanceSampleCount() override;
    void GenerateRays() override;
    void IntersectRays(std::uint32_t bounce) override;
    void ComputeAOVs() override;
    void ShadeMissedRays(std::uint32_t bounce) override;
    void ShadeSurfaceHits(std::uint32_t bounce) override;
    void IntersectShadowRays() override;
    void AccumulateDirectSamples() override;
    void ClearOutgoingRayCounter(std::uint32_t bounce) override;
    void ClearShadowRayCounter() override;
    void Denoise() override;
    void CopyHistoryBuffers() override;
    void ResolveRadiance() override;

private:
    void Check(int rc) const;
    void SyncOptions();

    HIPContext& context_;
    rt_frame* frame_ = nullptr;
    std::vector<float> resolved_;
    bool resolved_pinned_ = false;     // resolved_ is page-locked (rt_host_register)
    bool resolve_every_frame_ = true;
    std::string blue_noise_path_ = "assets/blue_noise/heitz2019_256spp_256d.bin";
};
} // namespace rt
