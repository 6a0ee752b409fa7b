#include "render.hpp"

namespace rt
{
Camera MakeCamera(float3 position, float yaw, float pitch, float fov, float aspect, float aperture, float focus)
{
    // camera_controller.cpp:77-80: front from yaw/pitch (Z-up), right = normalize(front x world_up), up = right x front
    float3 front(std::cos(yaw) * std::sin(pitch), std::sin(yaw) * std::sin(pitch), std::cos(pitch));
    float3 right = Cross(front, float3(0.0f, 0.0f, 1.0f)).Normalize();
    float3 up = Cross(right, front);
    Camera c = {};
    c.position = rt_float3{position.x, position.y, position.z, 0.0f};
    c.front = rt_float3{front.x, front.y, front.z, 0.0f};
    c.up = rt_float3{up.x, up.y, up.z, 0.0f};
    c.fov = fov;
    c.aspect_ratio = aspect;
    c.aperture = aperture;
    c.focus_distance = focus;
    return c;
}

Camera DefaultCamera(std::uint32_t width, std::uint32_t height)
{
    const float kPiDiv2 = 1.570796327f;                       // MATH_PIDIV2, mathlib.hpp:37
    return MakeCamera(float3(0.0f, -1.0f, 1.0f), kPiDiv2, kPiDiv2, 75.0f * 3.1415f / 180.0f,
        (float)width / (float)height, 0.0f, 10.0f);
}

Render::Render(std::uint32_t width, std::uint32_t height, Scene& scene, int device_ordinal, TileDesc tile)
    : scene_(scene), width_(width), height_(height)
{
    context_ = std::make_shared<HIPContext>(device_ordinal);
    // Build first, Finalize after: the build reorders the triangles the emissive
    // list indexes (render.cpp:61-67)
    auto bvh = std::make_unique<Bvh>();
    if (scene_.HasPrebuiltBvh()) bvh->AdoptNodes(scene_.GetPrebuiltNodes());   // binary scene cache
    else bvh->BuildCPU(scene_.GetTriangles());
    acc_structure_ = std::move(bvh);
    scene_.Finalize();
    integrator_ = std::make_unique<HIPPathTraceIntegrator>(width_, height_, *acc_structure_, *context_, tile);
    integrator_->UploadGPUData(scene_, *acc_structure_);
    camera_ = DefaultCamera(width_, height_);
}

void Render::SetCamera(Camera const& camera)
{
    camera_ = camera;
    camera_changed_ = true;
}

void Render::RenderFrame()
{
    integrator_->SetCameraData(camera_);
    if (camera_changed_)
    {
        integrator_->RequestReset();
        camera_changed_ = false;
    }
    integrator_->Integrate();
}

void Render::RenderSamples(std::uint32_t n)
{
    integrator_->SetCameraData(camera_);
    if (camera_changed_)
    {
        integrator_->RequestReset();
        camera_changed_ = false;
    }
    integrator_->IntegrateSamples(n);
}
} // namespace rt
