// integrator.cpp -- the per-sample stage schedule.  The ORDER of the hooks is the
// contract with backends and is the reference's (src/integrator/integrator.cpp:27-59);
// hip_pt_integrator.cpp documents which hooks the HIP backend fuses into no-ops.
#include "integrator.hpp"

namespace rt
{
Integrator::Integrator(std::uint32_t width, std::uint32_t height, AccelerationStructure& acc_structure)
    : width_(width), height_(height), acc_structure_(acc_structure)
{
}

// One bounce of the wavefront: closest hits, (AOVs on the first), escaped rays, queue
// counters, surface shading with NEE + BSDF sampling, shadow rays, direct light.
void Integrator::ScheduleBounce(std::uint32_t bounce)
{
    IntersectRays(bounce);
    if (bounce == 0)
        ComputeAOVs();
    ShadeMissedRays(bounce);
    ClearOutgoingRayCounter(bounce);
    ClearShadowRayCounter();
    ShadeSurfaceHits(bounce);
    IntersectShadowRays();
    AccumulateDirectSamples();
}

void Integrator::Integrate()
{
    // while denoising every frame restarts from an empty radiance buffer (the history
    // buffer carries the accumulation); otherwise only on request
    const bool restart = request_reset_ || enable_denoiser_;
    if (restart)
    {
        Reset();
        request_reset_ = false;
    }

    GenerateRays();
    // max_bounces_ is INCLUSIVE: B + 1 closest-hit passes and B + 1 shadow passes
    current_bounce_ = 0;
    do
    {
        ScheduleBounce(current_bounce_);
    } while (current_bounce_++ < max_bounces_);
    current_bounce_ = 0;

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
    request_reset_ = true;
}

void Integrator::EnableWhiteFurnace(bool enable)
{
    if (enable_white_furnace_ != enable)
    {
        enable_white_furnace_ = enable;
        CreateKernels();        // the furnace is a kernel variant (-D ENABLE_WHITE_FURNACE in the reference)
        request_reset_ = true;
    }
}
} // namespace rt
