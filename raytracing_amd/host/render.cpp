#include "render.hpp"
#include <chrono>
#include <thread>

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

Render::Render(std::uint32_t width, std::uint32_t height, Scene& scene, int device_ordinal, TileDesc tile,
    std::vector<std::pair<int, std::uint32_t>> const& context_options)
    : scene_(scene), width_(width), height_(height)
{
    context_ = std::make_shared<HIPContext>(device_ordinal);
    for (auto const& o : context_options)
        if (rt_ctx_set_option(context_->Get(), o.first, o.second) != RT_OK) throw HIPException(rt_last_error(context_->Get()));
    auto now = []() { return std::chrono::steady_clock::now(); };
    auto since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(now() - t).count(); };
    // Build first, Finalize after: the build reorders the triangles the emissive
    // list indexes (render.cpp:61-67)
    auto t = now();
    auto bvh = std::make_unique<Bvh>();
    if (scene_.HasPrebuiltBvh()) bvh->AdoptNodes(scene_.GetPrebuiltNodes());   // binary scene cache
    else bvh->BuildCPU(scene_.GetTriangles());
    acc_structure_ = std::move(bvh);
    setup_seconds_[0] = since(t); t = now();
    scene_.Finalize();
    setup_seconds_[1] = since(t); t = now();
    integrator_ = std::make_unique<HIPPathTraceIntegrator>(width_, height_, *acc_structure_, *context_, tile);
    setup_seconds_[2] = since(t); t = now();
    integrator_->UploadGPUData(scene_, *acc_structure_);
    setup_seconds_[3] = since(t);
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

TiledRender::TiledRender(std::uint32_t width, std::uint32_t height, Scene& scene, std::vector<int> const& devices,
    std::uint32_t band_height)
    : scene_(scene), width_(width), height_(height)
{
    if (devices.empty()) throw HIPException("TiledRender: no devices");
    auto bvh = std::make_unique<Bvh>();
    if (scene_.HasPrebuiltBvh()) bvh->AdoptNodes(scene_.GetPrebuiltNodes());
    else bvh->BuildCPU(scene_.GetTriangles());
    acc_structure_ = std::move(bvh);
    scene_.Finalize();
    for (std::size_t i = 0; i < devices.size(); ++i)
    {
        contexts_.push_back(std::make_unique<HIPContext>(devices[i]));
        // one adaptation per group (ShareFolds): tile 0 builds the shadow rays' tree and adapts its folds -- waited for by its first frame -- the others take its records
        if (devices.size() > 1)
        {
            if (i == 0) rt_ctx_set_option(contexts_[i]->Get(), RT_CTX_OPT_ADAPTIVE_FOLD, 25u | 2u);     // (the library's default + bit 1: its first frame waits for the adapted fold)
            else { rt_ctx_set_option(contexts_[i]->Get(), RT_CTX_OPT_SHADOW_TREE, 0u); rt_ctx_set_option(contexts_[i]->Get(), RT_CTX_OPT_ADAPTIVE_FOLD, 0u); }
        }
        TileDesc tile;
        tile.rank = (std::uint32_t)i;
        tile.count = (std::uint32_t)devices.size();
        tile.band_height = band_height;
        integrators_.push_back(std::make_unique<HIPPathTraceIntegrator>(width_, height_, *acc_structure_, *contexts_[i], tile));
        integrators_[i]->UploadGPUData(scene_, *acc_structure_);
        integrators_[i]->SetResolveEveryFrame(false);
    }
    bool shared = devices.size() > 1;
    for (int d : devices) shared = shared && d == devices[0];
    const int rc = shared ? rt_group_create_local((int)devices.size(), devices[0], &group_)
                          : rt_group_create((int)devices.size(), devices.data(), &group_);
    if (rc != RT_OK)
        throw HIPException(std::string("Failed to create the device group: ") + rt_group_last_error(nullptr));
    tile_seconds_.assign(devices.size(), 0.0);
    camera_ = DefaultCamera(width_, height_);
}

TiledRender::~TiledRender()
{
    rt_group_destroy(group_);
    integrators_.clear();            // frames before their contexts
    contexts_.clear();
}

std::vector<std::uint32_t> TiledRender::TileRows(std::uint32_t height, std::uint32_t rank, std::uint32_t count, std::uint32_t band_height)
{
    std::vector<std::uint32_t> rows;
    if (count == 0 || band_height == 0) return rows;
    for (std::uint64_t band = rank; band * band_height < height; band += count)
        for (std::uint64_t y = band * band_height; y < height && y < (band + 1) * band_height; ++y) rows.push_back((std::uint32_t)y);
    return rows;
}

int TiledRender::GetRcclRanks() const
{
    int n = 0;
    if (rt_group_comm_count(group_, 0, &n, nullptr) != RT_OK) throw HIPException(rt_group_last_error(group_));
    return n;
}

void TiledRender::SetCamera(Camera const& camera)
{
    camera_ = camera;
    camera_changed_ = true;
}

void TiledRender::SetMaxBounces(std::uint32_t max_bounces)
{
    for (auto& i : integrators_) i->SetMaxBounces(max_bounces);
}

void TiledRender::EnableWhiteFurnace(bool enable)
{
    for (auto& i : integrators_) i->EnableWhiteFurnace(enable);
}

void TiledRender::RenderSamples(std::uint32_t n)
{
    std::vector<std::thread> workers;
    std::vector<std::string> errors(integrators_.size());
    for (std::size_t i = 0; i < integrators_.size(); ++i)
        workers.emplace_back([&, i]()
        {
            try
            {
                auto t0 = std::chrono::steady_clock::now();
                integrators_[i]->SetCameraData(camera_);
                if (camera_changed_) integrators_[i]->RequestReset();
                integrators_[i]->IntegrateSamples(n);
                contexts_[i]->Finish();
                tile_seconds_[i] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            }
            catch (std::exception& ex) { errors[i] = ex.what(); }
        });
    for (auto& w : workers) w.join();
    camera_changed_ = false;
    for (auto& e : errors) if (!e.empty()) throw HIPException(e);
    if (!folds_shared_ && integrators_.size() > 1) ShareFolds();
}

void TiledRender::ShareFolds()
{
    folds_shared_ = true;
    rt_ctx* root = contexts_[0]->Get();
    std::uint32_t n = 0, m = 0, entries[2] = {0, 0};
    if (rt_scene_export_folds(root, nullptr, nullptr, 0, &n, &m, entries) != RT_OK) return;          // (no 4-wide tree: nothing to share)
    std::vector<unsigned char> cl((std::size_t)n * 64), sh((std::size_t)(m ? m : 1) * 64);
    if (rt_scene_export_folds(root, cl.data(), sh.data(), n > m ? n : m, &n, &m, entries) != RT_OK) throw HIPException(rt_last_error(root));
    for (std::size_t i = 1; i < contexts_.size(); ++i)
        if (rt_scene_import_folds(contexts_[i]->Get(), cl.data(), n, entries[0], m ? sh.data() : nullptr, m, entries[1]) != RT_OK)
            throw HIPException(rt_last_error(contexts_[i]->Get()));
}

std::vector<float> TiledRender::GatherRadiance(int root)
{
    std::vector<rt_frame*> frames;
    for (auto& i : integrators_) frames.push_back(i->GetFrame());
    std::vector<float> image((std::size_t)width_ * height_ * 4);
    if (rt_group_gather_radiance(group_, frames.data(), root, image.data(), nullptr) != RT_OK)
        throw HIPException(std::string("Radiance gather failed: ") + rt_group_last_error(group_));
    return image;
}

rt_stats TiledRender::GetStats() const
{
    rt_stats sum = {};
    for (auto& i : integrators_)
    {
        rt_stats st = i->GetStats();
        sum.closest_rays += st.closest_rays;
        sum.shadow_rays += st.shadow_rays;
        sum.samples = st.samples;
        sum.samples_in_flight = st.samples_in_flight;
        sum.path_state_bytes += st.path_state_bytes;
    }
    return sum;
}
} // namespace rt
