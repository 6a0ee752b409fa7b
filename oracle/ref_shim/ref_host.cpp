// ref_host.cpp -- TEST INFRASTRUCTURE.  C wrapper over the reference's own
// host-side producers, compiled from the sources where they lie under
// /root/reference (Scene: src/scene/scene.cpp, Bvh: src/bvh.cpp, LoadHDR:
// src/loaders/hdr_loader.cpp).  Used to pin this repo's Scene/Bvh/HDR
// restatements and to feed identical inputs to oracle, reference kernels and
// the HIP path.
#include "scene/scene.hpp"
#include "bvh.hpp"
#include "rt_types.h"
#include <vector>
#include <memory>

static_assert(sizeof(Triangle) == sizeof(rt_triangle), "Triangle");
static_assert(sizeof(LinearBVHNode) == sizeof(rt_bvh_node), "LinearBVHNode");
static_assert(sizeof(PackedMaterial) == sizeof(rt_packed_material), "PackedMaterial");
static_assert(sizeof(Light) == sizeof(rt_light), "Light");
static_assert(sizeof(Texture) == sizeof(rt_texture), "Texture");
static_assert(sizeof(Camera) == sizeof(rt_camera), "Camera");
static_assert(sizeof(SceneInfo) == sizeof(rt_scene_info), "SceneInfo");
static_assert(sizeof(Ray) == sizeof(rt_ray), "Ray");
static_assert(sizeof(Hit) == sizeof(rt_hit), "Hit");
static_assert(offsetof(Triangle, mtlIndex) == offsetof(rt_triangle, mtl_index), "mtlIndex");
static_assert(offsetof(LinearBVHNode, offset) == offsetof(rt_bvh_node, offset), "offset");
static_assert(offsetof(Camera, fov) == offsetof(rt_camera, fov), "fov");

struct RefScene
{
    std::unique_ptr<Scene> scene;
    std::unique_ptr<Bvh> bvh;
};

extern "C" {

void* refh_scene_load(const char* path, float scale, int flip_yz)
{
    try
    {
        auto* s = new RefScene;
        s->scene = std::make_unique<Scene>(path, scale, flip_yz != 0);
        return s;
    }
    catch (std::exception&)
    {
        return nullptr;
    }
}

void refh_scene_destroy(void* h) { delete (RefScene*)h; }

void refh_add_directional_light(void* h, float dx, float dy, float dz, float r, float g, float b)
{
    ((RefScene*)h)->scene->AddDirectionalLight(float3(dx, dy, dz), float3(r, g, b));
}

void refh_add_point_light(void* h, float x, float y, float z, float r, float g, float b)
{
    ((RefScene*)h)->scene->AddPointLight(float3(x, y, z), float3(r, g, b));
}

// Render::Render order (render.cpp:61-67): BuildCPU (reorders triangles), then Finalize()
void refh_build_and_finalize(void* h)
{
    auto* s = (RefScene*)h;
    s->bvh = std::make_unique<Bvh>();
    s->bvh->BuildCPU(s->scene->GetTriangles());
    s->scene->Finalize();   // loads assets/ibl/CGSkies_0036_free.hdr relative to CWD
}

uint32_t refh_num_triangles(void* h) { return (uint32_t)((RefScene*)h)->scene->GetTriangles().size(); }
const void* refh_triangles(void* h) { return ((RefScene*)h)->scene->GetTriangles().data(); }
uint32_t refh_num_nodes(void* h) { return (uint32_t)((RefScene*)h)->bvh->GetNodes().size(); }
const void* refh_nodes(void* h) { return ((RefScene*)h)->bvh->GetNodes().data(); }
uint32_t refh_num_materials(void* h) { return (uint32_t)((RefScene*)h)->scene->GetMaterials().size(); }
const void* refh_materials(void* h) { return ((RefScene*)h)->scene->GetMaterials().data(); }
uint32_t refh_num_textures(void* h) { return (uint32_t)((RefScene*)h)->scene->GetTextures().size(); }
const void* refh_textures(void* h) { return ((RefScene*)h)->scene->GetTextures().data(); }
uint32_t refh_num_texture_data(void* h) { return (uint32_t)((RefScene*)h)->scene->GetTextureData().size(); }
const void* refh_texture_data(void* h) { return ((RefScene*)h)->scene->GetTextureData().data(); }
uint32_t refh_num_lights(void* h) { return (uint32_t)((RefScene*)h)->scene->GetLights().size(); }
const void* refh_lights(void* h) { return ((RefScene*)h)->scene->GetLights().data(); }
uint32_t refh_num_emissive(void* h) { return (uint32_t)((RefScene*)h)->scene->GetEmissiveIndices().size(); }
const void* refh_emissive(void* h) { return ((RefScene*)h)->scene->GetEmissiveIndices().data(); }
void refh_scene_info(void* h, rt_scene_info* out)
{
    SceneInfo const& si = ((RefScene*)h)->scene->GetSceneInfo();
    memcpy(out, &si, sizeof(si));
}
uint32_t refh_env_width(void* h) { return ((RefScene*)h)->scene->GetEnvImage().width; }
uint32_t refh_env_height(void* h) { return ((RefScene*)h)->scene->GetEnvImage().height; }
const void* refh_env_data(void* h) { return ((RefScene*)h)->scene->GetEnvImage().data.data(); }

// Bvh::BuildCPU on a caller-supplied triangle array (reordered in place).
// Returns the node count; nodes_out may be NULL to query.
static std::unique_ptr<Bvh> g_last_bvh;
uint32_t refh_bvh_build(rt_triangle* tris, uint32_t n)
{
    std::vector<Triangle> v((Triangle*)tris, (Triangle*)tris + n);
    g_last_bvh = std::make_unique<Bvh>();
    g_last_bvh->BuildCPU(v);
    memcpy(tris, v.data(), sizeof(Triangle) * n);
    return (uint32_t)g_last_bvh->GetNodes().size();
}
void refh_bvh_nodes(rt_bvh_node* out)
{
    memcpy(out, g_last_bvh->GetNodes().data(), g_last_bvh->GetNodes().size() * sizeof(LinearBVHNode));
}

// LoadHDR (hdr_loader.cpp:29-100): returns 0 on failure; data = float RGBA
static Image g_img;
int refh_load_hdr(const char* path, uint32_t* w, uint32_t* h)
{
    g_img = Image();
    if (!LoadHDR(path, g_img)) return 0;
    *w = g_img.width; *h = g_img.height;
    return 1;
}
// LoadSTB (image_loader.cpp:30-63): stb_image decode -> packed RGBA8
int refh_load_stb(const char* path, uint32_t* w, uint32_t* h)
{
    g_img = Image();
    if (!LoadSTB(path, g_img)) return 0;
    *w = g_img.width; *h = g_img.height;
    return 1;
}
const void* refh_loaded_image_data() { return g_img.data.data(); }

} // extern "C"
