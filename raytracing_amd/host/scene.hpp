// scene.hpp -- the scene-loader surface the integrator consumes
// (reference: src/scene/scene.hpp:34-67 -- same constructor, getters, Finalize,
// AddPointLight, AddDirectionalLight).
#pragma once
#include <string>
#include <unordered_map>
#include <vector>
#include "structures.hpp"

namespace rt
{
bool LoadHDR(const char* filename, Image& result);   // Radiance RGBE (reference: src/loaders/hdr_loader.cpp:29-100)
bool LoadTGA(const char* filename, Image& result);
bool LoadPNG(const char* filename, Image& result);   // png_loader.cpp
bool LoadJPEG(const char* filename, Image& result);  // jpeg_loader.cpp (stb_image's integer pipeline, texel for texel)   // 8-bit TGA -> packed RGBA8 (reference path: LoadSTB, image_loader.cpp:30-63)

class Scene
{
public:
    // ---- construction -----------------------------------------------------------
    // OBJ + MTL from disk (main.cpp:56); throws std::runtime_error on failure.  A file written
    // by SaveCache is recognised by its magic and loaded as is (scale / flip_yz are baked in).
    Scene(const char* filename, float scale, bool flip_yz);
    // ... with opt-in extensions (not in the reference; include/rt_hip.h, rt_scene_desc):
    //   kWideTextureIndices: texture indices are kept 16 bits wide in a side table, lifting the 255-texture limit of the
    //                        packed material (PackAlbedo's assert, scene.cpp:55; constants.h:35) -- a scene with more
    //                        textures fails to load without it
    //   kEmissiveNee:        next-event estimation also samples the emissive triangles (GetEmissiveIndices)
    enum Options : unsigned { kWideTextureIndices = 1u, kEmissiveNee = 2u };
    Scene(const char* filename, float scale, bool flip_yz, unsigned options);
    // Caller-built arrays (procedural scenes / binary caches); no file IO.
    Scene(std::vector<Triangle> triangles, std::vector<PackedMaterial> materials, std::vector<Texture> textures,
        std::vector<std::uint32_t> texture_data);

    // ---- lights (OBJ has none; main.cpp:58 adds one directional light) -----------
    void AddPointLight(float3 origin, float3 radiance);
    void AddDirectionalLight(float3 direction, float3 radiance);   // stored as the unit vector TOWARDS the light

    // ---- finalisation: emissive list (after the BVH reorder), light count, env map.
    // The env map defaults to the path the reference hard-codes relative to the CWD
    // (scene.cpp:353-361); both setters are extensions for headless/batch use.
    void Finalize();
    void SetEnvironmentPath(std::string path) { env_path_ = std::move(path); }
    void SetEnvironmentImage(Image image) { env_image_ = std::move(image); env_preset_ = true; }

    // ---- binary cache (scene_cache.cpp): triangles in BVH order + nodes + materials + textures
    void SaveCache(const char* path, std::vector<LinearBVHNode> const& nodes) const;
    static bool IsCacheFile(const char* filename);
    bool HasPrebuiltBvh() const { return !prebuilt_nodes_.empty(); }
    std::vector<LinearBVHNode> const& GetPrebuiltNodes() const { return prebuilt_nodes_; }

    // ---- what the integrator uploads (scene.hpp:39-47) -----------------------------
    std::vector<Triangle>& GetTriangles() { return triangles_; }   // mutable: Bvh::BuildCPU reorders them
    std::vector<Triangle> const& GetTriangles() const { return triangles_; }
    std::vector<PackedMaterial> const& GetMaterials() const { return materials_; }
    std::vector<Texture> const& GetTextures() const { return textures_; }
    std::vector<std::uint32_t> const& GetTextureData() const { return texture_data_; }
    std::vector<Light> const& GetLights() const { return lights_; }
    std::vector<std::uint32_t> const& GetEmissiveIndices() const { return emissive_indices_; }
    SceneInfo const& GetSceneInfo() const { return scene_info_; }
    // extensions: 6 texture indices per material (diffuse, specular, roughness, metalness, emission, transparency;
    // 0xFFFF = none), empty unless kWideTextureIndices / SetMaterialTextureIndices; and the RT_SCENE_* flags
    std::vector<std::uint16_t> const& GetMaterialTextureIndices() const { return material_texture_indices_; }
    void SetMaterialTextureIndices(std::vector<std::uint16_t> indices);   // for caller-built arrays
    void SetEmissiveNee(bool enable) { emissive_nee_ = enable; }
    bool GetEmissiveNee() const { return emissive_nee_; }
    Image const& GetEnvImage() const { return env_image_; }

private:
    void Load(const char* filename, float scale, bool flip_yz);
    void LoadCache(const char* path);
    std::size_t LoadTexture(const std::string& filename);   // cached by file name, returns the texture index
    void CollectEmissiveTriangles();

    std::vector<Triangle> triangles_;
    std::vector<PackedMaterial> materials_;
    std::vector<Texture> textures_;
    std::vector<std::uint32_t> texture_data_;               // RGBA8 atlas, all textures back to back
    std::unordered_map<std::string, std::size_t> loaded_textures_;
    std::vector<Light> lights_;
    std::vector<std::uint32_t> emissive_indices_;
    std::vector<LinearBVHNode> prebuilt_nodes_;             // from a cache file: Bvh adopts them instead of building
    std::vector<std::uint16_t> material_texture_indices_;   // extension: see GetMaterialTextureIndices
    bool wide_texture_indices_ = false;
    bool emissive_nee_ = false;
    SceneInfo scene_info_ = {};
    Image env_image_;
    std::string env_path_ = "assets/ibl/CGSkies_0036_free.hdr";
    bool env_preset_ = false;
};
} // namespace rt
