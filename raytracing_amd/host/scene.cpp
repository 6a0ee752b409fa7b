// scene.cpp -- OBJ/MTL ingest and material packing for the HIP backend's host
// layer.  Behaviour follows the reference's src/scene/scene.cpp:53-361; the
// reference delegates file parsing to tinyobjloader v2.0.0 (vendored,
// 3rdparty/tinyobjloader/tiny_obj_loader.h), whose published behaviour is
// restated here for the subset of OBJ/MTL the path needs:
//   * number parsing = tinyobj's tryParseDouble (:837-965): digit-by-digit
//     mantissa accumulation in double, fractional digits scaled by a 1e-k table
//     then pow(10,-k), exponent via ldexp(m * 5^e, e); result narrowed to float
//     -- NOT strtod, so vertex positions match the reference bit for bit
//   * v / vt / vn / f (v, v/vt, v//vn, v/vt/vn; negative = relative indices),
//     usemtl, mtllib; o / g / s only delimit shapes and do not change triangle order
//   * triangles kept as is; quads split along the shorter diagonal (:1397-1500);
//     larger polygons are fanned (tinyobj ear-clips them -- documented deviation)
//   * MTL: newmtl Ka Kd Ks Ke Kt/Tf Ni Ns illum d Tr Pr Pm map_Kd map_Ks map_Pr
//     map_Pm map_Ke map_d, defaults of InitMaterial (:1303-1360)
#include "scene.hpp"
#include <cassert>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <cctype>
#include <dirent.h>
#include <sys/stat.h>

namespace rt
{
// ---------------------------------------------------------------------------
// packing helpers (scene.cpp:53-124)
// ---------------------------------------------------------------------------
namespace
{
unsigned PackAlbedo(float r, float g, float b, std::uint32_t texture_index)
{
    assert(texture_index < 256);
    r = clamp(r, 0.0f, 1.0f);
    g = clamp(g, 0.0f, 1.0f);
    b = clamp(b, 0.0f, 1.0f);
    return ((unsigned)(r * 255.0f)) | ((unsigned)(g * 255.0f) << 8) | ((unsigned)(b * 255.0f) << 16) |
           (texture_index << 24);
}

unsigned PackRGBE(float r, float g, float b)
{
    r = std::max(r, 0.0f);
    g = std::max(g, 0.0f);
    b = std::max(b, 0.0f);
    float v = r;
    if (g > v) v = g;
    if (b > v) v = b;
    if (v < 1e-32f) return 0;
    int e;
    v = std::frexp(v, &e) * 256.0f / v;
    return ((unsigned)(r * v)) | ((unsigned)(g * v) << 8) | ((unsigned)(b * v) << 16) | ((unsigned)(e + 128) << 24);
}

float3 UnpackRGBEHost(unsigned rgbe)
{
    int r = (rgbe >> 0) & 0xFF, g = (rgbe >> 8) & 0xFF, b = (rgbe >> 16) & 0xFF;
    int e = rgbe >> 24;
    if (!e) return float3(0.0f);
    float f = std::ldexp(1.0f, e - (int)(128 + 8));
    return float3((float)r * f, (float)g * f, (float)b * f);
}

unsigned PackRoughnessMetalness(float roughness, std::uint32_t ridx, float metalness, std::uint32_t midx)
{
    assert(ridx < 256 && midx < 256);
    roughness = clamp(roughness, 0.0f, 1.0f);
    metalness = clamp(metalness, 0.0f, 1.0f);
    return ((unsigned)(roughness * 255.0f)) | (ridx << 8) | ((unsigned)(metalness * 255.0f) << 16) | (midx << 24);
}

unsigned PackIorEmissionIdxTransparency(float ior, std::uint32_t eidx, float transparency, std::uint32_t tidx)
{
    assert(eidx < 256 && tidx < 256);
    ior = clamp(ior, 0.0f, 10.0f);
    transparency = clamp(transparency, 0.0f, 1.0f);
    return ((unsigned)(ior * 25.5f)) | (eidx << 8) | ((unsigned)(transparency * 255.0f) << 16) | (tidx << 24);
}

// ---------------------------------------------------------------------------
// text parsing
// ---------------------------------------------------------------------------
inline bool IsSpace(char c) { return c == ' ' || c == '\t'; }
inline bool IsDigit(char c) { return c >= '0' && c <= '9'; }

// tinyobj tryParseDouble (tiny_obj_loader.h:837-965)
bool ParseDouble(const char* s, const char* s_end, double* result)
{
    if (s >= s_end) return false;
    double mantissa = 0.0;
    int exponent = 0;
    char sign = '+', exp_sign = '+';
    const char* curr = s;
    int read = 0;
    bool end_not_reached = false;
    bool leading_dot = false;

    if (*curr == '+' || *curr == '-')
    {
        sign = *curr;
        curr++;
        if (curr != s_end && *curr == '.') leading_dot = true;
    }
    else if (IsDigit(*curr)) {}
    else if (*curr == '.') leading_dot = true;
    else return false;

    end_not_reached = (curr != s_end);
    if (!leading_dot)
    {
        while (end_not_reached && IsDigit(*curr))
        {
            mantissa *= 10;
            mantissa += (int)(*curr - '0');
            curr++;
            read++;
            end_not_reached = (curr != s_end);
        }
        if (read == 0) return false;
    }
    bool have_exp = false;
    if (end_not_reached)
    {
        if (*curr == '.')
        {
            curr++;
            read = 1;
            end_not_reached = (curr != s_end);
            static const double lut[] = {1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001};
            const int lut_n = sizeof lut / sizeof lut[0];
            while (end_not_reached && IsDigit(*curr))
            {
                mantissa += (int)(*curr - '0') * (read < lut_n ? lut[read] : std::pow(10.0, -read));
                read++;
                curr++;
                end_not_reached = (curr != s_end);
            }
            have_exp = end_not_reached && (*curr == 'e' || *curr == 'E');
        }
        else if (*curr == 'e' || *curr == 'E')
        {
            have_exp = true;
        }
    }
    if (have_exp)
    {
        curr++;
        end_not_reached = (curr != s_end);
        if (end_not_reached && (*curr == '+' || *curr == '-'))
        {
            exp_sign = *curr;
            curr++;
        }
        else if (end_not_reached && IsDigit(*curr)) {}
        else return false;
        read = 0;
        end_not_reached = (curr != s_end);
        while (end_not_reached && IsDigit(*curr))
        {
            if (exponent > std::numeric_limits<int>::max() / 10) return false;
            exponent *= 10;
            exponent += (int)(*curr - '0');
            curr++;
            read++;
            end_not_reached = (curr != s_end);
        }
        exponent *= (exp_sign == '+' ? 1 : -1);
        if (read == 0) return false;
    }
    *result = (sign == '+' ? 1 : -1) *
              (exponent ? std::ldexp(mantissa * std::pow(5.0, exponent), exponent) : mantissa);
    return true;
}

float ParseReal(const char** token, double def = 0.0)
{
    (*token) += strspn(*token, " \t");
    const char* end = (*token) + strcspn(*token, " \t\r");
    double val = def;
    ParseDouble(*token, end, &val);
    *token = end;
    return (float)val;
}

struct VertexIndex { int v = -1, vt = -1, vn = -1; };

// tinyobj fixIndex (:771-790): 1-based positive, negative relative to the current count
bool FixIndex(int idx, int n, int* ret)
{
    if (idx > 0) { *ret = idx - 1; return true; }
    if (idx == 0) return false;
    *ret = n + idx;
    return true;
}

// tinyobj parseTriple (:1105-1155)
bool ParseTriple(const char** token, int vsize, int vnsize, int vtsize, VertexIndex* ret)
{
    VertexIndex vi;
    if (!FixIndex(atoi(*token), vsize, &vi.v)) return false;
    (*token) += strcspn(*token, "/ \t\r");
    if ((*token)[0] != '/') { *ret = vi; return true; }
    (*token)++;
    if ((*token)[0] == '/')   // v//vn
    {
        (*token)++;
        if (!FixIndex(atoi(*token), vnsize, &vi.vn)) return false;
        (*token) += strcspn(*token, "/ \t\r");
        *ret = vi;
        return true;
    }
    if (!FixIndex(atoi(*token), vtsize, &vi.vt)) return false;   // v/vt[/vn]
    (*token) += strcspn(*token, "/ \t\r");
    if ((*token)[0] != '/') { *ret = vi; return true; }
    (*token)++;
    if (!FixIndex(atoi(*token), vnsize, &vi.vn)) return false;
    (*token) += strcspn(*token, "/ \t\r");
    *ret = vi;
    return true;
}

struct ObjMaterial
{
    std::string name;
    float diffuse[3] = {0, 0, 0}, specular[3] = {0, 0, 0}, transmittance[3] = {0, 0, 0}, emission[3] = {0, 0, 0};
    float ior = 1.0f, roughness = 0.0f, metallic = 0.0f;
    std::string diffuse_tex, specular_tex, roughness_tex, metallic_tex, emissive_tex, alpha_tex;
};

bool GetLine(std::istream& is, std::string& line)   // handles \n, \r\n and a missing final newline
{
    line.clear();
    if (!std::getline(is, line)) return false;
    while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
    return true;
}

// The file a material's map_* line names, as a Windows program would find it: the reference runs on Windows only
// (src/utils/window.cpp:30-35), where `textures\\Wall_Diffuse.PNG` and `Textures/wall_diffuse.png` are the same file, and the
// assets it is pointed at (Bistro) are authored there.  Backslashes become slashes; a component that does not exist as
// written is looked up ignoring case.  A name that exists as written is returned unchanged.
std::string ResolveAssetPath(const std::string& folder, std::string name)
{
    for (char& c : name) if (c == '\\') c = '/';
    std::string direct = folder + "/" + name;
    struct stat st;
    if (stat(direct.c_str(), &st) == 0) return direct;
    std::string cur = folder;
    size_t pos = 0;
    while (pos <= name.size())
    {
        size_t next = name.find('/', pos);
        std::string part = name.substr(pos, next == std::string::npos ? std::string::npos : next - pos);
        pos = next == std::string::npos ? name.size() + 1 : next + 1;
        if (part.empty() || part == ".") continue;
        std::string cand = cur + "/" + part;
        if (stat(cand.c_str(), &st) != 0)
        {
            std::string found;
            if (DIR* d = opendir(cur.c_str()))
            {
                while (dirent* e = readdir(d))
                {
                    std::string n = e->d_name;
                    if (n.size() != part.size()) continue;
                    bool same = true;
                    for (size_t i = 0; i < n.size() && same; ++i) same = std::tolower((unsigned char)n[i]) == std::tolower((unsigned char)part[i]);
                    if (same && (found.empty() || n < found)) found = n;      // deterministic when several names differ by case only
                }
                closedir(d);
            }
            if (found.empty()) return direct;                                 // not there: the caller reports the name as written
            cand = cur + "/" + found;
        }
        cur = cand;
    }
    return cur;
}

std::string TextureName(const char* token)
{
    // What tinyobjloader hands the reference for a map_* line (tiny_obj_loader.h:1191-1270, ParseTextureNameAndOption): options
    // first -- each followed by a fixed number of words: one for -blendu -blendv -clamp -boost -bm -type -texres -imfchan
    // -colorspace, two for -mm, THREE for -o -s -t (it reads three reals whatever follows: `-s 1 1 wall.png` eats the name) --
    // then the REST of the line is the file name, blanks included.
    std::string s(token);
    while (!s.empty() && (s.back() == ' ' || s.back() == '\t' || s.back() == '\r' || s.back() == '\n')) s.pop_back();
    size_t p = 0;
    auto skip_blanks = [&]() { while (p < s.size() && (s[p] == ' ' || s[p] == '\t')) ++p; };
    auto skip_word = [&]() { skip_blanks(); while (p < s.size() && s[p] != ' ' && s[p] != '\t') ++p; };
    static const struct { const char* name; int words; } kOptions[] = {
        {"-blendu", 1}, {"-blendv", 1}, {"-clamp", 1}, {"-boost", 1}, {"-bm", 1}, {"-type", 1}, {"-texres", 1}, {"-imfchan", 1},
        {"-colorspace", 1}, {"-mm", 2}, {"-o", 3}, {"-s", 3}, {"-t", 3}};
    for (;;)
    {
        skip_blanks();
        if (p >= s.size()) return std::string();
        bool option = false;
        for (const auto& o : kOptions)
        {
            const size_t n = strlen(o.name);
            if (s.compare(p, n, o.name) == 0 && p + n < s.size() && (s[p + n] == ' ' || s[p + n] == '\t'))
            {
                p += n;
                for (int w = 0; w < o.words; ++w) skip_word();
                option = true;
                break;
            }
        }
        if (!option) return s.substr(p);
    }
}

void LoadMtl(const std::string& path, std::vector<ObjMaterial>& materials, std::unordered_map<std::string, int>& map)
{
    std::ifstream is(path);
    if (!is) return;   // tinyobj only warns; faces then fall back to material 0 (scene.cpp:260-268)
    ObjMaterial m;
    bool has_kd = false;
    auto flush = [&]()
    {
        if (!m.name.empty())
        {
            map.insert(std::make_pair(m.name, (int)materials.size()));
            materials.push_back(m);
        }
    };
    std::string line;
    while (GetLine(is, line))
    {
        size_t e = line.find_last_not_of(" \t");
        line = line.substr(0, e == std::string::npos ? 0 : e + 1);
        const char* t = line.c_str();
        t += strspn(t, " \t");
        if (t[0] == '\0' || t[0] == '#') continue;
        auto real3 = [&](float* dst) { dst[0] = ParseReal(&t); dst[1] = ParseReal(&t); dst[2] = ParseReal(&t); };
        if (!strncmp(t, "newmtl", 6) && IsSpace(t[6]))
        {
            flush();
            m = ObjMaterial();
            has_kd = false;
            m.name = std::string(t + 7);
            continue;
        }
        if (t[0] == 'K' && t[1] == 'd' && IsSpace(t[2])) { t += 2; real3(m.diffuse); has_kd = true; continue; }
        if (t[0] == 'K' && t[1] == 's' && IsSpace(t[2])) { t += 2; real3(m.specular); continue; }
        if ((t[0] == 'K' && t[1] == 't' && IsSpace(t[2])) || (t[0] == 'T' && t[1] == 'f' && IsSpace(t[2])))
        {
            t += 2; real3(m.transmittance); continue;
        }
        if (t[0] == 'N' && t[1] == 'i' && IsSpace(t[2])) { t += 2; m.ior = ParseReal(&t); continue; }
        if (t[0] == 'K' && t[1] == 'e' && IsSpace(t[2])) { t += 2; real3(m.emission); continue; }
        if (t[0] == 'P' && t[1] == 'r' && IsSpace(t[2])) { t += 2; m.roughness = ParseReal(&t); continue; }
        if (t[0] == 'P' && t[1] == 'm' && IsSpace(t[2])) { t += 2; m.metallic = ParseReal(&t); continue; }
        if (!strncmp(t, "map_Kd", 6) && IsSpace(t[6]))
        {
            m.diffuse_tex = TextureName(t + 7);
            if (!has_kd) { m.diffuse[0] = m.diffuse[1] = m.diffuse[2] = 0.6f; }   // tinyobj default
            continue;
        }
        if (!strncmp(t, "map_Ks", 6) && IsSpace(t[6])) { m.specular_tex = TextureName(t + 7); continue; }
        if (!strncmp(t, "map_Pr", 6) && IsSpace(t[6])) { m.roughness_tex = TextureName(t + 7); continue; }
        if (!strncmp(t, "map_Pm", 6) && IsSpace(t[6])) { m.metallic_tex = TextureName(t + 7); continue; }
        if (!strncmp(t, "map_Ke", 6) && IsSpace(t[6])) { m.emissive_tex = TextureName(t + 7); continue; }
        if (!strncmp(t, "map_d", 5) && IsSpace(t[5])) { m.alpha_tex = TextureName(t + 6); continue; }
        // Ka Ns illum d Tr Ps Pc ... : parsed by tinyobj, unused by the path
    }
    flush();
}
} // namespace

// ---------------------------------------------------------------------------
// Scene
// ---------------------------------------------------------------------------
Scene::Scene(const char* filename, float scale, bool flip_yz) : Scene(filename, scale, flip_yz, 0u) {}

Scene::Scene(const char* filename, float scale, bool flip_yz, unsigned options)
{
    wide_texture_indices_ = (options & kWideTextureIndices) != 0;
    emissive_nee_ = (options & kEmissiveNee) != 0;
    if (IsCacheFile(filename))
    {
        LoadCache(filename);
        // a cache holds the scene as it was loaded: scale and axis flip were applied when it was written
        if (scale != 1.0f || flip_yz)
            std::fprintf(stderr, "warning: %s is a scene cache; --scale / --flip_yz were fixed when it was written and are ignored\n", filename);
    }
    else Load(filename, scale, flip_yz);
}

Scene::Scene(std::vector<Triangle> triangles, std::vector<PackedMaterial> materials, std::vector<Texture> textures,
    std::vector<std::uint32_t> texture_data)
    : triangles_(std::move(triangles)), materials_(std::move(materials)), textures_(std::move(textures)),
      texture_data_(std::move(texture_data))
{
}

void Scene::SetMaterialTextureIndices(std::vector<std::uint16_t> indices)
{
    if (!indices.empty() && indices.size() != materials_.size() * 6)
        throw std::runtime_error("SetMaterialTextureIndices: 6 entries per material expected");
    for (std::uint16_t t : indices)
        if (t != 0xFFFFu && t >= textures_.size()) throw std::runtime_error("SetMaterialTextureIndices: texture index out of range");
    material_texture_indices_ = std::move(indices);
    wide_texture_indices_ = !material_texture_indices_.empty();
}

void Scene::Load(const char* filename, float scale, bool flip_yz)
{
    std::string fname(filename);
    size_t slash = fname.find_last_of("/\\");
    std::string folder = slash == std::string::npos ? std::string() : fname.substr(0, slash);
    std::string mtl_base = folder.empty() ? std::string() : folder + "/";

    std::ifstream is(fname);
    if (!is) throw std::runtime_error("Failed to load the scene!");

    std::vector<float> v, vn, vt;
    std::vector<ObjMaterial> obj_materials;
    std::unordered_map<std::string, int> material_map;
    int material = -1;
    struct Face { VertexIndex i[3]; int material; };
    std::vector<Face> faces;

    std::string line;
    while (GetLine(is, line))
    {
        const char* t = line.c_str();
        t += strspn(t, " \t");
        if (t[0] == '\0' || t[0] == '#') continue;
        if (t[0] == 'v' && IsSpace(t[1]))
        {
            t += 2;
            v.push_back(ParseReal(&t)); v.push_back(ParseReal(&t)); v.push_back(ParseReal(&t));
            continue;
        }
        if (t[0] == 'v' && t[1] == 'n' && IsSpace(t[2]))
        {
            t += 3;
            vn.push_back(ParseReal(&t)); vn.push_back(ParseReal(&t)); vn.push_back(ParseReal(&t));
            continue;
        }
        if (t[0] == 'v' && t[1] == 't' && IsSpace(t[2]))
        {
            t += 3;
            vt.push_back(ParseReal(&t)); vt.push_back(ParseReal(&t));
            continue;
        }
        if (t[0] == 'f' && IsSpace(t[1]))
        {
            t += 2;
            t += strspn(t, " \t");
            std::vector<VertexIndex> poly;
            while (t[0] != '\0' && t[0] != '\r' && t[0] != '\n')
            {
                VertexIndex vi;
                if (!ParseTriple(&t, (int)(v.size() / 3), (int)(vn.size() / 3), (int)(vt.size() / 2), &vi))
                    throw std::runtime_error("Failed to load the scene!");
                poly.push_back(vi);
                t += strspn(t, " \t\r");
            }
            auto emit = [&](int a, int b, int c) { faces.push_back(Face{{poly[a], poly[b], poly[c]}, material}); };
            if (poly.size() == 3) emit(0, 1, 2);
            else if (poly.size() == 4)
            {
                // tinyobj quad rule: cut along the shorter diagonal (the positions must exist: a face may name a
                // vertex that is defined later or never -- the general index check comes after the parse)
                for (const VertexIndex& q : poly)
                    if (q.v < 0 || (size_t)q.v * 3 + 2 >= v.size()) throw std::runtime_error("Failed to load the scene!");
                auto P = [&](int k, int c) { return v[(size_t)poly[k].v * 3 + c]; };
                float e02x = P(2, 0) - P(0, 0), e02y = P(2, 1) - P(0, 1), e02z = P(2, 2) - P(0, 2);
                float e13x = P(3, 0) - P(1, 0), e13y = P(3, 1) - P(1, 1), e13z = P(3, 2) - P(1, 2);
                float sqr02 = e02x * e02x + e02y * e02y + e02z * e02z;
                float sqr13 = e13x * e13x + e13y * e13y + e13z * e13z;
                if (sqr02 < sqr13) { emit(0, 1, 2); emit(0, 2, 3); }
                else { emit(0, 1, 3); emit(1, 2, 3); }
            }
            else if (poly.size() > 4)
            {
                for (size_t k = 1; k + 1 < poly.size(); ++k) emit(0, (int)k, (int)k + 1);
            }
            continue;
        }
        if (!strncmp(t, "usemtl", 6) && IsSpace(t[6]))
        {
            t += 7;
            t += strspn(t, " \t");
            std::string name(t);
            while (!name.empty() && IsSpace(name.back())) name.pop_back();
            auto it = material_map.find(name);
            material = it == material_map.end() ? -1 : it->second;
            continue;
        }
        if (!strncmp(t, "mtllib", 6) && IsSpace(t[6]))
        {
            t += 7;
            std::stringstream ss(t);
            std::string lib;
            while (ss >> lib)
            {
                size_t before = obj_materials.size();
                LoadMtl(folder.empty() ? mtl_base + lib : ResolveAssetPath(folder, lib), obj_materials, material_map);
                if (obj_materials.size() > before) break;   // first library that loads wins
            }
            continue;
        }
        // o / g / s / others: no effect on the flattened triangle list
    }

    // materials (scene.cpp:145-186)
    materials_.resize(obj_materials.size());
    const float kGamma = 2.2f;
    const std::uint32_t kInvalidTextureIndex = 0xFF;
    // wide mode (extension): the index goes to the 16-bit side table and the packed 8-bit field says "none".  Asked for by
    // the caller (Scene::kWideTextureIndices) or switched on -- with a warning -- when the scene turns out to hold more
    // textures than the reference's 8-bit fields can name (Bistro does): the table is always filled, and adopted afterwards.
    std::vector<std::uint16_t> wide(obj_materials.size() * 6, 0xFFFFu);
    size_t wide_at = 0;
    auto tex = [&](const std::string& name) -> std::uint32_t
    {
        const size_t slot = wide_at++;
        if (name.empty()) return kInvalidTextureIndex;
        const std::uint32_t idx = (std::uint32_t)LoadTexture(ResolveAssetPath(folder, name));
        wide[slot] = (std::uint16_t)idx;
        return idx < kInvalidTextureIndex ? idx : kInvalidTextureIndex;
    };
    for (size_t i = 0; i < obj_materials.size(); ++i)
    {
        const ObjMaterial& in = obj_materials[i];
        PackedMaterial& out = materials_[i];
        wide_at = i * 6;      // slot order of rt_scene_desc::material_texture_indices: diffuse, specular, roughness, metalness, emission, transparency
        const std::uint32_t didx = tex(in.diffuse_tex);
        const std::uint32_t sidx = tex(in.specular_tex);
        out.diffuse_albedo = PackAlbedo(std::pow(in.diffuse[0], kGamma), std::pow(in.diffuse[1], kGamma),
            std::pow(in.diffuse[2], kGamma), didx);
        out.specular_albedo = PackAlbedo(std::pow(in.specular[0], kGamma), std::pow(in.specular[1], kGamma),
            std::pow(in.specular[2], kGamma), sidx);
        out.emission = PackRGBE(in.emission[0], in.emission[1], in.emission[2]);
        std::uint32_t ridx = tex(in.roughness_tex);
        std::uint32_t midx = tex(in.metallic_tex);
        out.roughness_metalness = PackRoughnessMetalness(in.roughness, ridx, in.metallic, midx);
        std::uint32_t eidx = tex(in.emissive_tex);
        std::uint32_t tidx = tex(in.alpha_tex);
        out.ior_emission_idx_transparency = PackIorEmissionIdxTransparency(in.ior, eidx, in.transmittance[0], tidx);
    }
    if (!wide_texture_indices_ && textures_.size() > 255)
    {
        std::fprintf(stderr, "warning: %s uses %zu textures, more than the 255 the reference's 8-bit texture indices can name "
                             "(constants.h:35, scene.cpp:55); switching to 16-bit texture indices (Scene::kWideTextureIndices)\n",
            filename, textures_.size());
        wide_texture_indices_ = true;
    }
    if (wide_texture_indices_)
    {
        for (PackedMaterial& m : materials_)           // the packed fields say "none": the side table names the textures
        {
            m.diffuse_albedo |= 0xFF000000u;
            m.specular_albedo |= 0xFF000000u;
            m.roughness_metalness |= 0xFF00FF00u;
            m.ior_emission_idx_transparency |= 0xFF00FF00u;
        }
        material_texture_indices_ = std::move(wide);
    }
    else material_texture_indices_.clear();

    // triangles (scene.cpp:188-270)
    auto flip = [flip_yz](float3& p)
    {
        if (flip_yz) { std::swap(p.y, p.z); p.y = -p.y; }
    };
    triangles_.reserve(faces.size());
    for (const Face& f : faces)
    {
        Vertex vx[3];
        for (int k = 0; k < 3; ++k)
        {
            const VertexIndex& ix = f.i[k];
            if (ix.v < 0 || (size_t)ix.v * 3 + 2 >= v.size()) throw std::runtime_error("Failed to load the scene!");
            vx[k].position = float3(v[(size_t)ix.v * 3 + 0] * scale, v[(size_t)ix.v * 3 + 1] * scale,
                v[(size_t)ix.v * 3 + 2] * scale);
            if (ix.vn >= 0 && (size_t)ix.vn * 3 + 2 < vn.size())
                vx[k].normal = float3(vn[(size_t)ix.vn * 3 + 0], vn[(size_t)ix.vn * 3 + 1], vn[(size_t)ix.vn * 3 + 2]);
            else
                vx[k].normal = float3(0.0f);   // the reference reads normals[-3..] here (UB); filled in below
            if (ix.vt >= 0 && (size_t)ix.vt * 2 + 1 < vt.size())
                vx[k].texcoord = float3(vt[(size_t)ix.vt * 2 + 0], vt[(size_t)ix.vt * 2 + 1], 0.0f);
        }
        for (int k = 0; k < 3; ++k)
        {
            if (f.i[k].vn < 0)   // no vn in the file: use the face normal
            {
                float3 n = Cross(vx[1].position - vx[0].position, vx[2].position - vx[0].position);
                float l = n.Length();
                vx[k].normal = l > 0.0f ? float3(n.x / l, n.y / l, n.z / l) : float3(0.0f, 0.0f, 1.0f);
            }
        }
        for (int k = 0; k < 3; ++k) { flip(vx[k].position); flip(vx[k].normal); }
        std::uint32_t mtl = (f.material >= 0 && (size_t)f.material < materials_.size()) ? (std::uint32_t)f.material : 0u;
        triangles_.emplace_back(vx[0], vx[1], vx[2], mtl);
    }
    if (materials_.empty()) materials_.push_back(PackedMaterial{0, 0, 0, 0, 0});
}

std::size_t Scene::LoadTexture(const std::string& filename)   // scene.cpp:276-322
{
    auto it = loaded_textures_.find(filename);
    if (it != loaded_textures_.end()) return it->second;
    size_t dot = filename.find_last_of('.');
    if (dot == std::string::npos) throw std::runtime_error("Invalid texture extension");
    std::string ext = filename.substr(dot);
    for (char& c : ext) c = (char)std::tolower((unsigned char)c);            // ".PNG": stb_image decides by content, the reference by strcmp
    Image image;
    bool ok = false;
    if (ext == ".tga") ok = LoadTGA(filename.c_str(), image);
    else if (ext == ".png") ok = LoadPNG(filename.c_str(), image);
    else if (ext == ".jpg") ok = LoadJPEG(filename.c_str(), image);
    if (!ok) throw std::runtime_error("Failed to load file " + filename);
    if (textures_.size() >= 0xFFFFu) throw std::runtime_error("More than 65535 textures");
    Texture t;
    t.width = (int)image.width;
    t.height = (int)image.height;
    t.data_start = (int)texture_data_.size();
    t.padding = 0;
    std::size_t idx = textures_.size();
    textures_.push_back(t);
    texture_data_.insert(texture_data_.end(), image.data.begin(), image.data.end());
    loaded_textures_.emplace(filename, idx);
    return idx;
}

void Scene::CollectEmissiveTriangles()   // scene.cpp:324-339
{
    emissive_indices_.clear();
    for (std::uint32_t i = 0; i < triangles_.size(); ++i)
    {
        float3 e = UnpackRGBEHost(materials_[triangles_[i].mtlIndex].emission);
        if (e.x + e.y + e.z > 0.0f) emissive_indices_.push_back(i);
    }
    scene_info_.emissive_count = (std::uint32_t)emissive_indices_.size();
}

void Scene::AddPointLight(float3 origin, float3 radiance)
{
    Light l = {};
    l.origin = rt_float3{origin.x, origin.y, origin.z, 0.0f};
    l.radiance = rt_float3{radiance.x, radiance.y, radiance.z, 0.0f};
    l.type = RT_LIGHT_TYPE_POINT;
    lights_.push_back(l);
}

void Scene::AddDirectionalLight(float3 direction, float3 radiance)   // stores the unit vector TOWARDS the light
{
    float3 d = direction.Normalize();
    Light l = {};
    l.origin = rt_float3{d.x, d.y, d.z, 0.0f};
    l.radiance = rt_float3{radiance.x, radiance.y, radiance.z, 0.0f};
    l.type = RT_LIGHT_TYPE_DIRECTIONAL;
    lights_.push_back(l);
}

void Scene::Finalize()   // scene.cpp:353-361
{
    CollectEmissiveTriangles();
    scene_info_.analytic_light_count = (std::uint32_t)lights_.size();
    if (!env_preset_)
    {
        if (!LoadHDR(env_path_.c_str(), env_image_))
            throw std::runtime_error("Failed to load the environment map " + env_path_);
    }
}

// ---------------------------------------------------------------------------
// Radiance .hdr (RGBE, new-style RLE) -> float RGBA, rows in file order, alpha 0
// (reference: src/loaders/hdr_loader.cpp:29-207)
// ---------------------------------------------------------------------------
namespace
{
bool OldDecrunch(unsigned char (*scan)[4], int len, FILE* f)
{
    int rshift = 0;
    unsigned char (*const first)[4] = scan;
    while (len > 0)
    {
        scan[0][0] = (unsigned char)fgetc(f);
        scan[0][1] = (unsigned char)fgetc(f);
        scan[0][2] = (unsigned char)fgetc(f);
        scan[0][3] = (unsigned char)fgetc(f);
        if (feof(f)) return false;
        if (scan[0][0] == 1 && scan[0][1] == 1 && scan[0][2] == 1)
        {
            if (scan == first) return false;                 // a run marker needs a previous pixel to repeat
            for (unsigned char i = (unsigned char)(scan[0][3] << rshift); i > 0 && len > 0; i--)   // never past the scanline
            {
                memcpy(&scan[0][0], &scan[-1][0], 4);
                scan++;
                len--;
            }
            rshift += 8;
        }
        else
        {
            scan++;
            len--;
            rshift = 0;
        }
    }
    return true;
}

bool Decrunch(unsigned char (*scan)[4], int len, FILE* f)
{
    if (len < 8 || len > 0x7fff) return OldDecrunch(scan, len, f);
    int i = fgetc(f);
    if (i != 2)
    {
        fseek(f, -1, SEEK_CUR);
        return OldDecrunch(scan, len, f);
    }
    scan[0][1] = (unsigned char)fgetc(f);
    scan[0][2] = (unsigned char)fgetc(f);
    i = fgetc(f);
    if (scan[0][1] != 2 || (scan[0][2] & 128))
    {
        scan[0][0] = 2;
        scan[0][3] = (unsigned char)i;
        return OldDecrunch(scan + 1, len - 1, f);
    }
    for (int c = 0; c < 4; c++)
    {
        for (int j = 0; j < len;)
        {
            unsigned char code = (unsigned char)fgetc(f);
            if (code > 128)
            {
                code &= 127;
                unsigned char val = (unsigned char)fgetc(f);
                while (code-- && j < len) scan[j++][c] = val;
            }
            else
            {
                while (code-- && j < len) scan[j++][c] = (unsigned char)fgetc(f);
            }
        }
    }
    return feof(f) ? false : true;
}
} // namespace

bool LoadHDR(const char* filename, Image& res)
{
    FILE* f = fopen(filename, "rb");
    if (!f) return false;
    char str[16];
    if (fread(str, 10, 1, f) != 1 || memcmp(str, "#?RADIANCE", 10)) { fclose(f); return false; }
    fseek(f, 1, SEEK_CUR);
    int c = 0, oldc;
    for (;;)   // header lines up to the blank line
    {
        oldc = c;
        c = fgetc(f);
        if (c == EOF) { fclose(f); return false; }
        if (c == 0xa && oldc == 0xa) break;
    }
    char reso[200];
    int i = 0;
    for (;;)
    {
        c = fgetc(f);
        if (c == EOF || i >= 199) { fclose(f); return false; }
        reso[i++] = (char)c;
        if (c == 0xa) break;
    }
    reso[i] = 0;
    long w = 0, h = 0;
    if (sscanf(reso, "-Y %ld +X %ld", &h, &w) != 2 || w <= 0 || h <= 0) { fclose(f); return false; }
    res.width = (std::uint32_t)w;
    res.height = (std::uint32_t)h;
    res.data.assign((size_t)w * h * 4, 0u);
    float* cols = (float*)res.data.data();
    std::vector<unsigned char> line((size_t)w * 4 + 4);
    auto scan = (unsigned char(*)[4])line.data();
    for (long y = h - 1; y >= 0; y--)
    {
        if (!Decrunch(scan, (int)w, f)) break;
        for (long x = 0; x < w; ++x)
        {
            int expo = scan[x][3] - 128;
            float d = std::ldexp(1.0f, expo);   // == powf(2.0f, expo), hdr_loader.cpp:102-107
            cols[0] = (scan[x][0] / 256.0f) * d;
            cols[1] = (scan[x][1] / 256.0f) * d;
            cols[2] = (scan[x][2] / 256.0f) * d;
            cols += 4;
        }
    }
    fclose(f);
    return true;
}

// ---------------------------------------------------------------------------
// TGA (types 2/3/10/11, 8/24/32 bpp) -> r | g<<8 | b<<16 | a<<24, top row first
// (what stbi_load + LoadSTB produce, image_loader.cpp:30-63; alpha 0 when absent)
// ---------------------------------------------------------------------------
bool LoadTGA(const char* filename, Image& res)
{
    FILE* f = fopen(filename, "rb");
    if (!f) return false;
    unsigned char hd[18];
    if (fread(hd, 18, 1, f) != 1) { fclose(f); return false; }
    int id_len = hd[0], cmap = hd[1], type = hd[2];
    int w = hd[12] | (hd[13] << 8), h = hd[14] | (hd[15] << 8), bpp = hd[16], desc = hd[17];
    bool rle = type == 10 || type == 11;
    bool grey = type == 3 || type == 11;
    if (cmap != 0 || !(type == 2 || type == 3 || rle) || w <= 0 || h <= 0) { fclose(f); return false; }
    int bytes = bpp / 8;
    if (!((grey && bytes == 1) || (!grey && (bytes == 3 || bytes == 4)))) { fclose(f); return false; }
    fseek(f, id_len, SEEK_CUR);
    std::vector<unsigned char> px((size_t)w * h * bytes);
    if (!rle)
    {
        if (fread(px.data(), px.size(), 1, f) != 1) { fclose(f); return false; }
    }
    else
    {
        size_t n = (size_t)w * h, i = 0;
        while (i < n)
        {
            int c = fgetc(f);
            if (c == EOF) { fclose(f); return false; }
            int cnt = (c & 127) + 1;
            if (c & 128)
            {
                unsigned char v[4];
                if (fread(v, bytes, 1, f) != 1) { fclose(f); return false; }
                for (int k = 0; k < cnt && i < n; ++k, ++i) memcpy(&px[i * bytes], v, bytes);
            }
            else
            {
                for (int k = 0; k < cnt && i < n; ++k, ++i)
                    if (fread(&px[i * bytes], bytes, 1, f) != 1) { fclose(f); return false; }
            }
        }
    }
    fclose(f);
    res.width = (std::uint32_t)w;
    res.height = (std::uint32_t)h;
    res.data.resize((size_t)w * h);
    bool top_origin = (desc & 0x20) != 0;
    for (int y = 0; y < h; ++y)
    {
        int sy = top_origin ? y : h - 1 - y;
        for (int x = 0; x < w; ++x)
        {
            const unsigned char* p = &px[((size_t)sy * w + x) * bytes];
            std::uint32_t r, g, b, a;
            if (grey) { r = p[0]; g = 0; b = 0; a = 0; }                    // 1 channel: LoadSTB keeps only r
            else { r = p[2]; g = p[1]; b = p[0]; a = bytes == 4 ? p[3] : 0; }
            res.data[(size_t)y * w + x] = r | (g << 8) | (b << 16) | (a << 24);
        }
    }
    return true;
}
} // namespace rt
