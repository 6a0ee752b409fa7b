// scene_cache.cpp -- binary scene cache: everything Scene::Load and
// Bvh::BuildCPU produce -- the REORDERED triangles, the LinearBVHNode array, packed
// materials, texture table and RGBA8 atlas -- in one file, so that a multi-million-triangle
// scene skips the OBJ/MTL parse, the texture decodes and the BVH build at start-up.
// Lights and the environment map are not part of it: the reference adds them after loading
// (main.cpp:58, scene.cpp:353-361) and so does the caller here.
//
// Layout (little endian): 64-byte header, then the five arrays back to back.
//   magic "RTSCENE\1" | u32 version | u32 sizeof(Triangle) | u32 sizeof(LinearBVHNode) |
//   u32 sizeof(PackedMaterial) | u64 counts[5] (triangles, nodes, materials, textures,
//   texture words) | u64 checksum of the payload
// Version 2 (written only when the scene carries the wide texture indices of Scene::kWideTextureIndices) appends
// u64 count + that many uint16 (6 per material) after the five arrays; version 1 files stay byte-identical.
#include "scene.hpp"
#include <cstdio>
#include <cstring>
#include <stdexcept>

namespace rt
{
namespace
{
constexpr char kMagic[8] = {'R', 'T', 'S', 'C', 'E', 'N', 'E', 1};
constexpr std::uint32_t kVersion = 1, kVersionWideTextures = 2;

struct Header
{
    char magic[8];
    std::uint32_t version, triangle_size, node_size, material_size;
    std::uint64_t counts[5];
    std::uint64_t checksum;
};
static_assert(sizeof(Header) == 72, "cache header layout");

// order-sensitive 64-bit checksum, 8 bytes per step (fast enough to stay IO-bound)
struct Checksum
{
    std::uint64_t h = 0x9E3779B97F4A7C15ull;
    void Add(const void* data, std::size_t bytes)
    {
        const unsigned char* p = static_cast<const unsigned char*>(data);
        std::size_t words = bytes / 8;
        for (std::size_t i = 0; i < words; ++i)
        {
            std::uint64_t w;
            std::memcpy(&w, p + i * 8, 8);
            h = (h ^ w) * 0xD6E8FEB86659FD93ull;
            h = (h << 31) | (h >> 33);
        }
        std::uint64_t tail = 0;
        std::memcpy(&tail, p + words * 8, bytes - words * 8);
        h = (h ^ tail ^ (std::uint64_t)bytes) * 0xD6E8FEB86659FD93ull;
    }
};

struct File
{
    std::FILE* f = nullptr;
    File(const char* path, const char* mode) : f(std::fopen(path, mode)) {}
    ~File() { if (f) std::fclose(f); }
};

template <class T>
void WriteArray(std::FILE* f, std::vector<T> const& v, Checksum& sum, const char* path)
{
    if (!v.empty() && std::fwrite(v.data(), sizeof(T), v.size(), f) != v.size())
        throw std::runtime_error(std::string("scene cache: short write to ") + path);
    sum.Add(v.data(), v.size() * sizeof(T));
}

template <class T>
void ReadArray(std::FILE* f, std::vector<T>& v, std::uint64_t count, Checksum& sum, const char* path)
{
    v.resize((std::size_t)count);
    if (count && std::fread(v.data(), sizeof(T), (std::size_t)count, f) != count)
        throw std::runtime_error(std::string("scene cache: truncated file ") + path);
    sum.Add(v.data(), v.size() * sizeof(T));
}
} // namespace

bool Scene::IsCacheFile(const char* filename)
{
    File in(filename, "rb");
    char magic[8];
    return in.f && std::fread(magic, 1, 8, in.f) == 8 && std::memcmp(magic, kMagic, 8) == 0;
}

void Scene::SaveCache(const char* path, std::vector<LinearBVHNode> const& nodes) const
{
    if (nodes.empty()) throw std::runtime_error("scene cache: no BVH to save (build it first)");
    File out(path, "wb");
    if (!out.f) throw std::runtime_error(std::string("scene cache: cannot create ") + path);
    Header h = {};
    std::memcpy(h.magic, kMagic, 8);
    h.version = material_texture_indices_.empty() ? kVersion : kVersionWideTextures;
    h.triangle_size = sizeof(Triangle); h.node_size = sizeof(LinearBVHNode); h.material_size = sizeof(PackedMaterial);
    h.counts[0] = triangles_.size(); h.counts[1] = nodes.size(); h.counts[2] = materials_.size();
    h.counts[3] = textures_.size(); h.counts[4] = texture_data_.size();
    if (std::fwrite(&h, sizeof(h), 1, out.f) != 1) throw std::runtime_error(std::string("scene cache: short write to ") + path);
    Checksum sum;
    WriteArray(out.f, triangles_, sum, path);
    WriteArray(out.f, nodes, sum, path);
    WriteArray(out.f, materials_, sum, path);
    WriteArray(out.f, textures_, sum, path);
    WriteArray(out.f, texture_data_, sum, path);
    if (h.version == kVersionWideTextures)
    {
        const std::uint64_t n = material_texture_indices_.size();
        if (std::fwrite(&n, sizeof(n), 1, out.f) != 1) throw std::runtime_error(std::string("scene cache: short write to ") + path);
        sum.Add(&n, sizeof(n));
        WriteArray(out.f, material_texture_indices_, sum, path);
    }
    h.checksum = sum.h;
    if (std::fseek(out.f, 0, SEEK_SET) != 0 || std::fwrite(&h, sizeof(h), 1, out.f) != 1)
        throw std::runtime_error(std::string("scene cache: cannot finish ") + path);
}

void Scene::LoadCache(const char* path)
{
    File in(path, "rb");
    if (!in.f) throw std::runtime_error(std::string("scene cache: cannot open ") + path);
    Header h;
    if (std::fread(&h, sizeof(h), 1, in.f) != 1 || std::memcmp(h.magic, kMagic, 8) != 0)
        throw std::runtime_error(std::string("scene cache: not a scene cache: ") + path);
    if ((h.version != kVersion && h.version != kVersionWideTextures) || h.triangle_size != sizeof(Triangle) || h.node_size != sizeof(LinearBVHNode) ||
        h.material_size != sizeof(PackedMaterial))
        throw std::runtime_error(std::string("scene cache: written by an incompatible version: ") + path);
    if (h.counts[0] == 0 || h.counts[1] == 0 || h.counts[0] > 0xFFFFFFFFull || h.counts[1] > 0xFFFFFFFFull)
        throw std::runtime_error(std::string("scene cache: implausible counts in ") + path);
    Checksum sum;
    ReadArray(in.f, triangles_, h.counts[0], sum, path);
    ReadArray(in.f, prebuilt_nodes_, h.counts[1], sum, path);
    ReadArray(in.f, materials_, h.counts[2], sum, path);
    ReadArray(in.f, textures_, h.counts[3], sum, path);
    ReadArray(in.f, texture_data_, h.counts[4], sum, path);
    material_texture_indices_.clear();
    if (h.version == kVersionWideTextures)
    {
        std::uint64_t n = 0;
        if (std::fread(&n, sizeof(n), 1, in.f) != 1 || n != h.counts[2] * 6)
            throw std::runtime_error(std::string("scene cache: truncated or inconsistent texture index table in ") + path);
        sum.Add(&n, sizeof(n));
        ReadArray(in.f, material_texture_indices_, n, sum, path);
    }
    wide_texture_indices_ = !material_texture_indices_.empty();
    if (sum.h != h.checksum) throw std::runtime_error(std::string("scene cache: checksum mismatch (corrupt file) ") + path);
    // the checksum covers accidents, not a stale or hand-made file: Finalize() and the upload index these arrays on the host
    auto bad = [&](const char* what) { return std::runtime_error(std::string("scene cache: inconsistent contents (") + what + ") in " + path); };
    for (const Triangle& t : triangles_)
        if (t.mtlIndex >= materials_.size()) throw bad("triangle material index");
    for (std::size_t i = 0; i < prebuilt_nodes_.size(); ++i)
    {
        const LinearBVHNode& n = prebuilt_nodes_[i];
        const std::uint32_t count = n.num_primitives_axis >> 16;
        if (count != 0) { if ((std::uint64_t)n.offset + count > triangles_.size()) throw bad("leaf range"); }
        else if (i + 1 >= prebuilt_nodes_.size() || n.offset <= i + 1 || n.offset >= prebuilt_nodes_.size() ||
                 (n.num_primitives_axis & 0xFFFFu) > 2u) throw bad("child offset / split axis");
    }
    for (const Texture& t : textures_)
        if (t.width <= 0 || t.height <= 0 || t.data_start < 0 ||
            (std::uint64_t)t.data_start + (std::uint64_t)t.width * (std::uint64_t)t.height > texture_data_.size()) throw bad("texture range");
    for (const PackedMaterial& m : materials_)
    {
        const std::uint32_t idx[6] = {m.diffuse_albedo >> 24, m.specular_albedo >> 24, (m.roughness_metalness >> 8) & 0xFFu,
            m.roughness_metalness >> 24, (m.ior_emission_idx_transparency >> 8) & 0xFFu, m.ior_emission_idx_transparency >> 24};
        if (material_texture_indices_.empty())
            for (std::uint32_t t : idx)
                if (t != 0xFFu && t >= textures_.size()) throw bad("material texture index");
    }
    for (std::uint16_t t : material_texture_indices_)
        if (t != 0xFFFFu && t >= textures_.size()) throw bad("wide material texture index");
}
} // namespace rt
