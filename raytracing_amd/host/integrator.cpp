// integrator.cpp -- the per-sample stage schedule (reference:
// src/integrator/integrator.cpp:27-77).  The order of the hooks is part of the
// contract with backends and is kept exactly; see hip_pt_integrator.cpp for
// which hooks the HIP backend fuses.
#include "integrator.hpp"

namespace rt
{
void Integrator::Integrate()
{
    if (request_reset_ || enable_denoiser_)
    {
        Reset();
        request_reset_ = false;
    }

    GenerateRays();

    // bounce runs 0..max_bounces_ INCLUSIVE: B + 1 closest-hit and shadow passes
    for (current_bounce_ = 0; current_bounce_ <= max_bounces_; ++current_bounce_)
    {
        const std::uint32_t bounce = current_bounce_;
        IntersectRays(bounce);
        if (bounce == 0) ComputeAOVs();
        ShadeMissedRays(bounce);
        ClearOutgoingRayCounter(bounce);
        ClearShadowRayCounter();
        ShadeSurfaceHits(bounce);
        IntersectShadowRays();
        AccumulateDirectSamples();
    }

    AdvanceSampleCount();
    if (enable_denoiser_)
    {
        Denoise();
        CopyHistoryBuffers();
    }
    ResolveRadiance();
}

void Integrator::SetMaxBounces(std::uint32_t max_bounces)
{
    max_bounces_ = max_bounces;
    RequestReset();
}

void Integrator::EnableWhiteFurnace(bool enable)
{
    if (enable == enable_white_furnace_) return;
    enable_white_furnace_ = enable;
    CreateKernels();
    RequestReset();
}
} // namespace rt
