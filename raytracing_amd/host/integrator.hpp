// integrator.hpp -- backend-agnostic wavefront schedule (upper drop-in boundary).
//
// Public surface and stage hooks are those of the reference's `class Integrator`
// (src/integrator/integrator.hpp:34-100): Render and GUI code written against the
// reference keep compiling against this class.  Differences: it does not include
// gpu_wrappers/cl_context.hpp (OpenCL + GL headers), it lives in namespace rt, and
// it records the bounce being scheduled for backends that key state by bounce.
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
    enum class SamplerType { kRandom, kBlueNoise };                           // integrator.hpp:37-41
    enum AOV { kShadedColor, kDiffuseAlbedo, kDepth, kNormal, kMotionVectors };   // integrator.hpp:43-50

    Integrator(std::uint32_t width, std::uint32_t height, AccelerationStructure& acc_structure);
    virtual ~Integrator() = default;

    // ---- per frame ------------------------------------------------------------
    void Integrate();                                  // one sample per pixel (integrator.cpp:27-59)
    void RequestReset() { request_reset_ = true; }     // restart accumulation at the next Integrate()

    // ---- scene / camera ---------------------------------------------------------
    virtual void UploadGPUData(Scene const& scene, AccelerationStructure const& acc_structure) = 0;
    virtual void SetCameraData(Camera const& camera) = 0;

    // ---- settings (each one requests a reset) -----------------------------------
    void SetMaxBounces(std::uint32_t max_bounces);     // bounce loop runs 0..max_bounces inclusive
    void EnableWhiteFurnace(bool enable);              // albedo 1, no emission, sky 0.5
    virtual void SetSamplerType(SamplerType sampler_type) = 0;
    virtual void SetAOV(AOV aov) = 0;
    virtual void EnableDenoiser(bool enable) = 0;

    std::uint32_t GetMaxBounces() const { return max_bounces_; }
    std::uint32_t GetWidth() const { return width_; }
    std::uint32_t GetHeight() const { return height_; }

protected:
    // ---- the fifteen stage hooks Integrate() schedules (integrator.hpp:65-79) ----
    virtual void CreateKernels() = 0;                                  // (re)select kernel variants
    virtual void Reset() = 0;                                          // zero radiance (+ sample counter)
    virtual void GenerateRays() = 0;                                   // RayGeneration
    virtual void IntersectRays(std::uint32_t bounce) = 0;              // TraceBvh
    virtual void ComputeAOVs() = 0;                                    // GenerateAOV, bounce 0 only
    virtual void ShadeMissedRays(std::uint32_t bounce) = 0;            // Miss
    virtual void ClearOutgoingRayCounter(std::uint32_t bounce) = 0;    // ClearCounter
    virtual void ClearShadowRayCounter() = 0;                          // ClearCounter
    virtual void ShadeSurfaceHits(std::uint32_t bounce) = 0;           // HitSurface
    virtual void IntersectShadowRays() = 0;                            // TraceBvh -D SHADOW_RAYS
    virtual void AccumulateDirectSamples() = 0;                        // AccumulateDirectSamples
    virtual void AdvanceSampleCount() = 0;                             // IncrementCounter
    virtual void Denoise() = 0;                                        // TemporalAccumulation
    virtual void CopyHistoryBuffers() = 0;                             // radiance/depth -> history
    virtual void ResolveRadiance() = 0;                                // ResolveRadiance

    std::uint32_t width_, height_;                     // render size
    AccelerationStructure& acc_structure_;
    Camera camera_ = {}, prev_camera_ = {};
    std::uint32_t max_bounces_ = 3u;                   // reference default (integrator.hpp:91)
    std::uint32_t current_bounce_ = 0;                 // bounce whose stages are being issued
    SamplerType sampler_type_ = SamplerType::kRandom;
    AOV aov_ = AOV::kShadedColor;
    bool request_reset_ = false;
    bool enable_white_furnace_ = false;                // debugging aid of the reference GUI
    bool enable_denoiser_ = false;

private:
    void ScheduleBounce(std::uint32_t bounce);
};
} // namespace rt
