// render.hpp -- headless frame orchestration: owns the HIP context, the BVH,
// the integrator and the camera (reference: src/render.{hpp,cpp}, minus the
// window, GL framebuffer and ImGui).  RenderFrame() = one sample per pixel.
#pragma once
#include <memory>
#include "bvh.hpp"
#include "hip_pt_integrator.hpp"
#include "scene.hpp"

namespace rt
{
// The reference's start-up camera (src/utils/camera_controller.cpp:30-41,77-80).
Camera DefaultCamera(std::uint32_t width, std::uint32_t height);
Camera MakeCamera(float3 position, float yaw, float pitch, float fov, float aspect, float aperture, float focus);

class Render
{
public:
    Render(std::uint32_t width, std::uint32_t height, Scene& scene, int device_ordinal = 0, TileDesc tile = TileDesc());

    void RenderFrame();                          // render.cpp:172-204 without present/GUI
    void RenderSamples(std::uint32_t n);         // n samples through the fused fast path
    void SetCamera(Camera const& camera);
    HIPPathTraceIntegrator& GetIntegrator() { return *integrator_; }
    HIPContext& GetContext() { return *context_; }
    AccelerationStructure const& GetAccelerationStructure() const { return *acc_structure_; }
    std::uint32_t GetWidth() const { return width_; }
    std::uint32_t GetHeight() const { return height_; }

private:
    Scene& scene_;
    std::uint32_t width_, height_;
    std::shared_ptr<HIPContext> context_;
    std::unique_ptr<AccelerationStructure> acc_structure_;
    std::unique_ptr<HIPPathTraceIntegrator> integrator_;
    Camera camera_;
    bool camera_changed_ = true;
};
} // namespace rt
