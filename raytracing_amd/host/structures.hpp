// structures.hpp -- C++ views of the shared host<->device records.  Layouts are
// the PODs of include/rt_types.h (reference src/kernels/common/shared_structures.h);
// these wrappers only add the constructors/helpers host code wants.
#pragma once
#include <vector>
#include "mathlib.hpp"
#include "rt_types.h"

namespace rt
{
struct Vertex
{
    float3 position, texcoord, normal;
    Vertex() = default;
    Vertex(const float3& p, const float2& uv, const float3& n) : position(p), texcoord(uv.x, uv.y, 0.0f), normal(n) {}
};

struct Triangle
{
    Vertex v1, v2, v3;
    std::uint32_t mtlIndex = 0;
    std::uint32_t padding[3] = {0, 0, 0};

    Triangle() = default;
    Triangle(const Vertex& a, const Vertex& b, const Vertex& c, std::uint32_t mtl) : v1(a), v2(b), v3(c), mtlIndex(mtl) {}
    Bounds3 GetBounds() const { return Union(Bounds3(v1.position, v2.position), v3.position); }
};

struct LinearBVHNode
{
    Bounds3 bounds;
    std::uint32_t offset = 0;               // first primitive (leaf) or second child (interior)
    std::uint32_t num_primitives_axis = 0;  // (n << 16) | axis ; n == 0 -> interior
    std::uint32_t padding[2] = {0, 0};
};

using PackedMaterial = rt_packed_material;
using Light = rt_light;
using Texture = rt_texture;
using SceneInfo = rt_scene_info;
using Camera = rt_camera;

static_assert(sizeof(Vertex) == sizeof(rt_vertex), "Vertex");
static_assert(sizeof(Triangle) == sizeof(rt_triangle), "Triangle");
static_assert(sizeof(LinearBVHNode) == sizeof(rt_bvh_node), "LinearBVHNode");

struct Image
{
    std::uint32_t width = 0, height = 0;
    std::vector<std::uint32_t> data;   // RGBA8 packed texels, or float RGBA bit patterns (HDR)
};
} // namespace rt
