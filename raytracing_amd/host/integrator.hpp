// integrator.hpp -- backend-agnostic wavefront schedule.
//
// This is the upper drop-in boundary: the public surface and the fifteen
// protected stage hooks are those of the reference's class Integrator
// (src/integrator/integrator.hpp:34-100) so that Render and any GUI code keep
// compiling against it; the only change is that it no longer drags in
// gpu_wrappers/cl_context.hpp (OpenCL + GL headers).
#pragma once
#include <cstdint>
#include "structures.hpp"

namespace rt
{
class Scene;
class AccelerationStructure;

class Integrator
{
public:
    enum class SamplerType { kRandom, kBlueNoise };
    enum AOV { kShadedColor, kDiffuseAlbedo, kDepth, kNormal, kMotionVectors };

    Integrator(std::uint32_t width, std::uint32_t height, AccelerationStructure& acc_structure)
        : width_(width), height_(height), acc_structure_(acc_structure) {}
    virtual ~Integrator() = default;

    // One sample per pixel: the stage sequence of integrator.cpp:27-59.
    void Integrate();
    virtual void UploadGPUData(Scene const& scene, AccelerationStructure const& acc_structure) = 0;
    virtual void SetCameraData(Camera const& camera) = 0;
    void RequestReset() { request_reset_ = true; }
    void EnableWhiteFurnace(bool enable);
    void SetMaxBounces(std::uint32_t max_bounces);
    virtual void SetSamplerType(SamplerType sampler_type) = 0;
    virtual void SetAOV(AOV aov) = 0;
    virtual void EnableDenoiser(bool enable) = 0;

    std::uint32_t GetMaxBounces() const { return max_bounces_; }

protected:
    virtual void CreateKernels() = 0;
    virtual void Reset() = 0;
    virtual void AdvanceSampleCount() = 0;
    virtual void GenerateRays() = 0;
    virtual void IntersectRays(std::uint32_t bounce) = 0;
    virtual void ComputeAOVs() = 0;
    virtual void ShadeMissedRays(std::uint32_t bounce) = 0;
    virtual void ShadeSurfaceHits(std::uint32_t bounce) = 0;
    virtual void IntersectShadowRays() = 0;
    virtual void AccumulateDirectSamples() = 0;
    virtual void ClearOutgoingRayCounter(std::uint32_t bounce) = 0;
    virtual void ClearShadowRayCounter() = 0;
    virtual void Denoise() = 0;
    virtual void CopyHistoryBuffers() = 0;
    virtual void ResolveRadiance() = 0;

    std::uint32_t width_;
    std::uint32_t height_;
    AccelerationStructure& acc_structure_;
    Camera camera_ = {};
    Camera prev_camera_ = {};
    std::uint32_t max_bounces_ = 3u;
    SamplerType sampler_type_ = SamplerType::kRandom;
    AOV aov_ = AOV::kShadedColor;
    bool request_reset_ = false;
    bool enable_white_furnace_ = false;
    bool enable_denoiser_ = false;
    // bounce currently being scheduled (IntersectShadowRays() takes no argument
    // in the reference API; the HIP backend keys its counters by bounce)
    std::uint32_t current_bounce_ = 0;
};
} // namespace rt
