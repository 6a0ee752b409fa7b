// host_capi.cpp -- flat C entry points over the C++ host layer (Scene, Bvh,
// Render, HIPPathTraceIntegrator) so that scripts and tests can drive it
// through ctypes.  Errors: functions return NULL / non-zero and the message is
// kept per thread (rth_last_error).
#include <cstring>
#include <string>
#include "render.hpp"

namespace
{
thread_local std::string g_err;
template <class F>
auto guard(F&& f, decltype(f()) fail_value) -> decltype(f())
{
    try { return f(); }
    catch (std::exception& e) { g_err = e.what(); return fail_value; }
}
} // namespace

extern "C" {

const char* rth_last_error() { return g_err.c_str(); }

// ---- Scene ----------------------------------------------------------------
void* rth_scene_load(const char* path, float scale, int flip_yz)
{
    return guard([&]() -> void* { return new rt::Scene(path, scale, flip_yz != 0); }, nullptr);
}

// options: rt::Scene::Options (1 = wide texture indices, 2 = emissive-triangle next-event estimation)
void* rth_scene_load_ex(const char* path, float scale, int flip_yz, unsigned options)
{
    return guard([&]() -> void* { return new rt::Scene(path, scale, flip_yz != 0, options); }, nullptr);
}
int rth_scene_set_material_texture_indices(void* s, const uint16_t* idx, uint32_t n)
{
    return guard([&]() { ((rt::Scene*)s)->SetMaterialTextureIndices(std::vector<uint16_t>(idx, idx + n)); return 0; }, 1);
}
void rth_scene_set_emissive_nee(void* s, int enable) { ((rt::Scene*)s)->SetEmissiveNee(enable != 0); }
int rth_scene_emissive_nee(void* s) { return ((rt::Scene*)s)->GetEmissiveNee() ? 1 : 0; }
uint32_t rth_scene_num_material_texture_indices(void* s) { return (uint32_t)((rt::Scene*)s)->GetMaterialTextureIndices().size(); }
const void* rth_scene_material_texture_indices(void* s) { return ((rt::Scene*)s)->GetMaterialTextureIndices().data(); }

void* rth_scene_from_arrays(const rt_triangle* tris, uint32_t ntris, const rt_packed_material* mats, uint32_t nmats,
    const rt_texture* tex, uint32_t ntex, const uint32_t* texdata, uint32_t ntexdata)
{
    return guard([&]() -> void*
    {
        std::vector<rt::Triangle> t(ntris);
        if (ntris) memcpy(static_cast<void*>(t.data()), tris, (size_t)ntris * sizeof(rt_triangle));
        std::vector<rt::PackedMaterial> m(mats, mats + nmats);
        std::vector<rt::Texture> tx(tex, tex + ntex);
        std::vector<uint32_t> td(texdata, texdata + ntexdata);
        return new rt::Scene(std::move(t), std::move(m), std::move(tx), std::move(td));
    }, nullptr);
}

void rth_scene_destroy(void* s) { delete (rt::Scene*)s; }
void rth_scene_add_directional_light(void* s, float dx, float dy, float dz, float r, float g, float b)
{
    ((rt::Scene*)s)->AddDirectionalLight(rt::float3(dx, dy, dz), rt::float3(r, g, b));
}
void rth_scene_add_point_light(void* s, float x, float y, float z, float r, float g, float b)
{
    ((rt::Scene*)s)->AddPointLight(rt::float3(x, y, z), rt::float3(r, g, b));
}
void rth_scene_set_env_path(void* s, const char* path) { ((rt::Scene*)s)->SetEnvironmentPath(path); }
int rth_scene_set_env_image(void* s, const float* rgba, uint32_t w, uint32_t h)
{
    rt::Image img;
    img.width = w; img.height = h;
    img.data.resize((size_t)w * h * 4);
    memcpy(img.data.data(), rgba, (size_t)w * h * 16);
    ((rt::Scene*)s)->SetEnvironmentImage(std::move(img));
    return 0;
}
int rth_scene_finalize(void* s) { return guard([&]() { ((rt::Scene*)s)->Finalize(); return 0; }, 1); }

uint32_t rth_scene_num_triangles(void* s) { return (uint32_t)((rt::Scene*)s)->GetTriangles().size(); }
const void* rth_scene_triangles(void* s) { return ((rt::Scene*)s)->GetTriangles().data(); }
uint32_t rth_scene_num_materials(void* s) { return (uint32_t)((rt::Scene*)s)->GetMaterials().size(); }
const void* rth_scene_materials(void* s) { return ((rt::Scene*)s)->GetMaterials().data(); }
uint32_t rth_scene_num_textures(void* s) { return (uint32_t)((rt::Scene*)s)->GetTextures().size(); }
const void* rth_scene_textures(void* s) { return ((rt::Scene*)s)->GetTextures().data(); }
uint32_t rth_scene_num_texture_data(void* s) { return (uint32_t)((rt::Scene*)s)->GetTextureData().size(); }
const void* rth_scene_texture_data(void* s) { return ((rt::Scene*)s)->GetTextureData().data(); }
uint32_t rth_scene_num_lights(void* s) { return (uint32_t)((rt::Scene*)s)->GetLights().size(); }
const void* rth_scene_lights(void* s) { return ((rt::Scene*)s)->GetLights().data(); }
uint32_t rth_scene_num_emissive(void* s) { return (uint32_t)((rt::Scene*)s)->GetEmissiveIndices().size(); }
const void* rth_scene_emissive(void* s) { return ((rt::Scene*)s)->GetEmissiveIndices().data(); }
uint32_t rth_scene_env_width(void* s) { return ((rt::Scene*)s)->GetEnvImage().width; }
uint32_t rth_scene_env_height(void* s) { return ((rt::Scene*)s)->GetEnvImage().height; }
const void* rth_scene_env_data(void* s) { return ((rt::Scene*)s)->GetEnvImage().data.data(); }

// ---- Bvh ------------------------------------------------------------------
void* rth_bvh_build(void* scene)
{
    return guard([&]() -> void*
    {
        auto* b = new rt::Bvh();
        auto* s = (rt::Scene*)scene;
        if (s->HasPrebuiltBvh()) b->AdoptNodes(s->GetPrebuiltNodes());
        else b->BuildCPU(s->GetTriangles());
        return b;
    }, nullptr);
}
void rth_bvh_destroy(void* b) { delete (rt::Bvh*)b; }
int rth_scene_save_cache(void* scene, void* bvh, const char* path)
{
    return guard([&]() { ((rt::Scene*)scene)->SaveCache(path, ((rt::Bvh*)bvh)->GetNodes()); return 0; }, 1);
}
uint32_t rth_bvh_num_nodes(void* b) { return (uint32_t)((rt::Bvh*)b)->GetNodes().size(); }
const void* rth_bvh_nodes(void* b) { return ((rt::Bvh*)b)->GetNodes().data(); }

// ---- loaders / camera -------------------------------------------------------
static rt::Image g_img;
int rth_load_hdr(const char* path, uint32_t* w, uint32_t* h)
{
    g_img = rt::Image();
    if (!rt::LoadHDR(path, g_img)) return 1;
    *w = g_img.width; *h = g_img.height;
    return 0;
}
int rth_load_jpeg(const char* path, uint32_t* w, uint32_t* h)
{
    g_img = rt::Image();
    if (!rt::LoadJPEG(path, g_img)) return 1;
    *w = g_img.width; *h = g_img.height;
    return 0;
}
int rth_load_tga(const char* path, uint32_t* w, uint32_t* h)
{
    g_img = rt::Image();
    if (!rt::LoadTGA(path, g_img)) return 1;
    *w = g_img.width; *h = g_img.height;
    return 0;
}
int rth_load_png(const char* path, uint32_t* w, uint32_t* h)
{
    g_img = rt::Image();
    if (!rt::LoadPNG(path, g_img)) return 1;
    *w = g_img.width; *h = g_img.height;
    return 0;
}
const void* rth_loaded_image_data() { return g_img.data.data(); }
void rth_default_camera(uint32_t w, uint32_t h, rt_camera* out) { *out = rt::DefaultCamera(w, h); }
void rth_make_camera(float px, float py, float pz, float yaw, float pitch, float fov, float aspect, float aperture,
    float focus, rt_camera* out)
{
    *out = rt::MakeCamera(rt::float3(px, py, pz), yaw, pitch, fov, aspect, aperture, focus);
}

// ---- Render (needs a GPU) ---------------------------------------------------
void* rth_render_create(uint32_t w, uint32_t h, void* scene, int device, uint32_t tile_rank, uint32_t tile_count,
    uint32_t band_height)
{
    return guard([&]() -> void*
    {
        rt::TileDesc t; t.rank = tile_rank; t.count = tile_count; t.band_height = band_height;
        return new rt::Render(w, h, *(rt::Scene*)scene, device, t);
    }, nullptr);
}
// ... with context options set before the upload: options = n_options x {option, value}
void* rth_render_create_with_options(uint32_t w, uint32_t h, void* scene, int device, uint32_t tile_rank, uint32_t tile_count,
    uint32_t band_height, const uint32_t* options, uint32_t n_options)
{
    return guard([&]() -> void*
    {
        rt::TileDesc t; t.rank = tile_rank; t.count = tile_count; t.band_height = band_height;
        std::vector<std::pair<int, std::uint32_t>> opts;
        for (uint32_t i = 0; i < n_options; ++i) opts.emplace_back((int)options[2 * i], options[2 * i + 1]);
        return new rt::Render(w, h, *(rt::Scene*)scene, device, t, opts);
    }, nullptr);
}
void rth_render_destroy(void* r) { delete (rt::Render*)r; }
int rth_render_set_camera(void* r, const rt_camera* cam) { return guard([&]() { ((rt::Render*)r)->SetCamera(*cam); return 0; }, 1); }
int rth_render_set_max_bounces(void* r, uint32_t b) { return guard([&]() { ((rt::Render*)r)->GetIntegrator().SetMaxBounces(b); return 0; }, 1); }
int rth_render_enable_white_furnace(void* r, int e) { return guard([&]() { ((rt::Render*)r)->GetIntegrator().EnableWhiteFurnace(e != 0); return 0; }, 1); }
int rth_render_set_sampler(void* r, int blue_noise)
{
    return guard([&]() { ((rt::Render*)r)->GetIntegrator().SetSamplerType(blue_noise ? rt::Integrator::SamplerType::kBlueNoise : rt::Integrator::SamplerType::kRandom); return 0; }, 1);
}
int rth_render_set_blue_noise_path(void* r, const char* path) { ((rt::Render*)r)->GetIntegrator().SetBlueNoiseTablePath(path); return 0; }
int rth_render_enable_denoiser(void* r, int e) { return guard([&]() { ((rt::Render*)r)->GetIntegrator().EnableDenoiser(e != 0); return 0; }, 1); }
int rth_render_set_aov(void* r, int aov) { return guard([&]() { ((rt::Render*)r)->GetIntegrator().SetAOV((rt::Integrator::AOV)aov); return 0; }, 1); }
int rth_render_resolve(void* r, float* out)
{
    return guard([&]()
    {
        auto const& v = ((rt::Render*)r)->GetIntegrator().ResolveNow();
        memcpy(out, v.data(), v.size() * sizeof(float));
        return 0;
    }, 1);
}
int rth_render_set_resolve_every_frame(void* r, int e) { ((rt::Render*)r)->GetIntegrator().SetResolveEveryFrame(e != 0); return 0; }
int rth_render_frame(void* r) { return guard([&]() { ((rt::Render*)r)->RenderFrame(); return 0; }, 1); }
int rth_render_samples(void* r, uint32_t n) { return guard([&]() { ((rt::Render*)r)->RenderSamples(n); return 0; }, 1); }
int rth_render_reserve_samples(void* r, uint32_t n)
{
    return guard([&]() { return (int)((rt::Render*)r)->GetIntegrator().ReserveSamples(n); }, -1);
}
int rth_render_finish(void* r) { return guard([&]() { ((rt::Render*)r)->GetContext().Finish(); return 0; }, 1); }
uint32_t rth_render_local_rows(void* r) { return ((rt::Render*)r)->GetIntegrator().GetLocalRows(); }
void rth_render_setup_seconds(void* r, double* out4) { const double* s = ((rt::Render*)r)->GetSetupSeconds(); for (int i = 0; i < 4; ++i) out4[i] = s[i]; }
uint32_t rth_render_global_row(void* r, uint32_t row) { return ((rt::Render*)r)->GetIntegrator().GetGlobalRow(row); }
uint32_t rth_render_sample_count(void* r) { return ((rt::Render*)r)->GetIntegrator().GetSampleCount(); }
int rth_render_read_radiance(void* r, float* out)
{
    return guard([&]()
    {
        std::vector<float> v = ((rt::Render*)r)->GetIntegrator().ReadRadianceSum();
        memcpy(out, v.data(), v.size() * sizeof(float));
        return 0;
    }, 1);
}
int rth_render_read_resolved(void* r, float* out)
{
    auto const& v = ((rt::Render*)r)->GetIntegrator().GetResolvedImage();
    memcpy(out, v.data(), v.size() * sizeof(float));
    return 0;
}
int rth_render_stats(void* r, rt_stats* out) { return guard([&]() { *out = ((rt::Render*)r)->GetIntegrator().GetStats(); return 0; }, 1); }
uint32_t rth_render_num_nodes(void* r) { return (uint32_t)((rt::Render*)r)->GetAccelerationStructure().GetNodes().size(); }
const void* rth_render_nodes(void* r) { return ((rt::Render*)r)->GetAccelerationStructure().GetNodes().data(); }
void* rth_render_frame_handle(void* r) { return ((rt::Render*)r)->GetIntegrator().GetFrame(); }
void* rth_render_ctx_handle(void* r) { return ((rt::Render*)r)->GetContext().Get(); }
int rth_render_upload_gpu_data(void* r) { return guard([&]() { ((rt::Render*)r)->UploadGPUData(); return 0; }, 1); }

} // extern "C"
