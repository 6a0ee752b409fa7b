// render.hpp -- headless frame orchestration: owns the HIP context, the BVH,
// the integrator and the camera (reference: src/render.{hpp,cpp}, minus the
// window, GL framebuffer and ImGui).  RenderFrame() = one sample per pixel.
#pragma once
#include <utility>
#include <memory>
#include <vector>
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
    // context_options: (rt_ctx_option, value) pairs set on the context before the scene is uploaded -- e.g. a rank of a group that will take another
    // rank's folds (rt_scene_import_folds) uploads without a shadow tree and without an adaptation of its own: {{2, 0}, {4, 0}}
    Render(std::uint32_t width, std::uint32_t height, Scene& scene, int device_ordinal = 0, TileDesc tile = TileDesc(),
        std::vector<std::pair<int, std::uint32_t>> const& context_options = {});

    void RenderFrame();                          // render.cpp:172-204 without present/GUI
    void RenderSamples(std::uint32_t n);         // n samples through the fused fast path
    void SetCamera(Camera const& camera);
    HIPPathTraceIntegrator& GetIntegrator() { return *integrator_; }
    HIPContext& GetContext() { return *context_; }
    // Uploads the scene again (after an rt_ctx_set_option that changes the device-side layout: tools, A/B runs)
    void UploadGPUData() { integrator_->UploadGPUData(scene_, *acc_structure_); }
    AccelerationStructure const& GetAccelerationStructure() const { return *acc_structure_; }
    std::uint32_t GetWidth() const { return width_; }
    std::uint32_t GetHeight() const { return height_; }
    // what the constructor spent, seconds: {Bvh::BuildCPU (or adopting a cached tree), Scene::Finalize, the integrator (frame buffers), UploadGPUData}
    double const* GetSetupSeconds() const { return setup_seconds_; }

private:
    Scene& scene_;
    std::uint32_t width_, height_;
    double setup_seconds_[4] = {0.0, 0.0, 0.0, 0.0};
    std::shared_ptr<HIPContext> context_;
    std::unique_ptr<AccelerationStructure> acc_structure_;
    std::unique_ptr<HIPPathTraceIntegrator> integrator_;
    Camera camera_;
    bool camera_changed_ = true;
};

// One image over several GPUs of this process (no reference counterpart: the reference drives
// devices_[0] only, src/gpu_wrappers/cl_context.cpp:89).  The scene and its BVH are built once and
// uploaded to every device; device i renders the interleaved row bands of tile i
// (rt_frame_desc) on its own host thread; GatherRadiance() is the one RCCL gather (rt_group_*).
class TiledRender
{
public:
    // devices[i] = the GPU of tile i.  All entries equal = every tile on ONE device, exchanged by device copies
    // (rt_group_create_local: RCCL refuses two ranks per GPU) -- the way to run the whole tiled path on a one-GPU box.
    TiledRender(std::uint32_t width, std::uint32_t height, Scene& scene, std::vector<int> const& devices,
        std::uint32_t band_height = 8);
    // The image rows tile `rank` of `count` owns: bands of band_height rows dealt round-robin (rt_frame_desc); no device needed.
    static std::vector<std::uint32_t> TileRows(std::uint32_t height, std::uint32_t rank, std::uint32_t count, std::uint32_t band_height = 8);
    int GetRcclRanks() const;                             // ncclCommCount of the group's communicator (0: local group)
    ~TiledRender();
    void SetCamera(Camera const& camera);
    void SetMaxBounces(std::uint32_t max_bounces);
    void EnableWhiteFurnace(bool enable);
    void RenderSamples(std::uint32_t n);                  // every tile, concurrently; returns when all are enqueued and finished
    std::vector<float> GatherRadiance(int root = 0);      // height x width x RGBA running sums, image order
    rt_stats GetStats() const;                            // ray counters summed over the tiles
    std::size_t GetTileCount() const { return integrators_.size(); }
    std::vector<double> const& GetLastTileSeconds() const { return tile_seconds_; }
    // One fold adaptation per GROUP: tile 0 adapts (its first RenderSamples waits for it), the other tiles -- uploaded without a shadow tree or an adaptation
    // of their own -- take its records (rt_scene_export_folds / rt_scene_import_folds).  Called by the first RenderSamples; results do not depend on it.
    void ShareFolds();
    bool FoldsShared() const { return folds_shared_; }
    HIPContext& GetContext(std::size_t i) { return *contexts_[i]; }
    AccelerationStructure const& GetAccelerationStructure() const { return *acc_structure_; }

private:
    Scene& scene_;
    std::uint32_t width_, height_;
    std::unique_ptr<AccelerationStructure> acc_structure_;
    std::vector<std::unique_ptr<HIPContext>> contexts_;
    std::vector<std::unique_ptr<HIPPathTraceIntegrator>> integrators_;
    std::vector<double> tile_seconds_;
    bool folds_shared_ = false;
    rt_group* group_ = nullptr;
    Camera camera_;
    bool camera_changed_ = true;
};
} // namespace rt
