// bvh.cpp -- PBRT-style binned SAH build, restated from the reference's
// src/bvh.cpp:67-245 so that, on the same input, the node array AND the
// reordered triangle array are identical (tests/test_host_pin.py checks this
// against the compiled reference):
//   * 12 buckets along the axis of maximum centroid extent (:139-150)
//   * split cost 1 + (n0*SA0 + n1*SA1)/SA (:153-169), first minimum wins (:172-181)
//   * leaf when n == 1, when all centroids coincide on that axis (:112), or when
//     n <= 4 and the best split costs at least n (:184-185)
//   * n == 2 -> median split with std::nth_element (:126-135)
//   * partition with std::partition, bucket <= best (:187-195); the relative
//     order inside each side is whatever libstdc++'s algorithm leaves, which
//     is why the same standard algorithms are used here
//   * depth-first flatten: first child = next node, second child index in
//     `offset`, leaves carry (n << 16) (:223-245)
// Differences from the reference: build nodes live in one vector instead of
// one `new` per node, recursion carries indices instead of pointers.
#include "bvh.hpp"
#include <cassert>
#include <iostream>

namespace rt
{
namespace
{
constexpr unsigned kMaxPrimsInLeaf = 4;
constexpr unsigned kBuckets = 12;

struct PrimInfo
{
    unsigned index = 0;
    Bounds3 bounds;
    float3 centroid;
};

struct BuildNode
{
    Bounds3 bounds;
    int child[2] = {-1, -1};
    int axis = 0, first = 0, count = 0;
};

struct Builder
{
    std::vector<Triangle> const& tris;
    std::vector<PrimInfo>& prims;
    std::vector<Triangle>& ordered;
    std::vector<BuildNode> pool;

    int MakeLeaf(int id, unsigned start, unsigned end, const Bounds3& bounds)
    {
        int first = (int)ordered.size();
        for (unsigned i = start; i < end; ++i) ordered.push_back(tris[prims[i].index]);
        pool[id].first = first;
        pool[id].count = (int)(end - start);
        pool[id].bounds = bounds;
        return id;
    }

    int Build(unsigned start, unsigned end)
    {
        int id = (int)pool.size();
        pool.emplace_back();

        Bounds3 bounds;
        for (unsigned i = start; i < end; ++i) bounds = Union(bounds, prims[i].bounds);

        unsigned n = end - start;
        if (n == 1) return MakeLeaf(id, start, end, bounds);

        Bounds3 cb;
        for (unsigned i = start; i < end; ++i) cb = Union(cb, prims[i].centroid);
        unsigned dim = cb.MaximumExtent();
        if (cb.max[dim] == cb.min[dim]) return MakeLeaf(id, start, end, bounds);

        unsigned mid = (start + end) / 2;
        if (n <= 2)
        {
            std::nth_element(&prims[start], &prims[mid], &prims[end - 1] + 1,
                [dim](const PrimInfo& a, const PrimInfo& b) { return a.centroid[dim] < b.centroid[dim]; });
        }
        else
        {
            int counts[kBuckets] = {0};
            Bounds3 bbounds[kBuckets];
            for (unsigned i = start; i < end; ++i)
            {
                int b = kBuckets * cb.Offset(prims[i].centroid)[dim];
                if (b == (int)kBuckets) b = kBuckets - 1;
                counts[b]++;
                bbounds[b] = Union(bbounds[b], prims[i].bounds);
            }
            float cost[kBuckets - 1];
            for (unsigned i = 0; i < kBuckets - 1; ++i)
            {
                Bounds3 b0, b1;
                int c0 = 0, c1 = 0;
                for (unsigned j = 0; j <= i; ++j) { b0 = Union(b0, bbounds[j]); c0 += counts[j]; }
                for (unsigned j = i + 1; j < kBuckets; ++j) { b1 = Union(b1, bbounds[j]); c1 += counts[j]; }
                cost[i] = 1.0f + (c0 * b0.SurfaceArea() + c1 * b1.SurfaceArea()) / bounds.SurfaceArea();
            }
            float best = cost[0];
            unsigned best_bucket = 0;
            for (unsigned i = 1; i < kBuckets - 1; ++i)
                if (cost[i] < best) { best = cost[i]; best_bucket = i; }

            float leaf_cost = float(n);
            if (n > kMaxPrimsInLeaf || best < leaf_cost)
            {
                PrimInfo* pmid = std::partition(&prims[start], &prims[end - 1] + 1, [=](const PrimInfo& pi)
                {
                    int b = kBuckets * cb.Offset(pi.centroid)[dim];
                    if (b == (int)kBuckets) b = kBuckets - 1;
                    return b <= (int)best_bucket;
                });
                mid = (unsigned)(pmid - &prims[0]);
            }
            else
            {
                return MakeLeaf(id, start, end, bounds);
            }
        }
        int c0 = Build(start, mid);
        int c1 = Build(mid, end);
        pool[id].child[0] = c0;
        pool[id].child[1] = c1;
        pool[id].bounds = Union(pool[c0].bounds, pool[c1].bounds);
        pool[id].axis = (int)dim;
        pool[id].count = 0;
        return id;
    }
};

unsigned Flatten(const std::vector<BuildNode>& pool, int id, std::vector<LinearBVHNode>& out, unsigned* offset)
{
    const BuildNode& n = pool[id];
    unsigned my = (*offset)++;
    out[my].bounds = n.bounds;
    if (n.count > 0)
    {
        assert(n.count < 65536);
        out[my].offset = (std::uint32_t)n.first;
        out[my].num_primitives_axis = (std::uint32_t)n.count << 16;
    }
    else
    {
        out[my].num_primitives_axis = (std::uint32_t)n.axis;
        Flatten(pool, n.child[0], out, offset);
        out[my].offset = Flatten(pool, n.child[1], out, offset);
    }
    return my;
}
} // namespace

void Bvh::BuildCPU(std::vector<Triangle>& triangles)
{
    if (verbose) std::cout << "Building Bounding Volume Hierarchy for scene" << std::endl;
    nodes_.clear();
    if (triangles.empty()) return;
    std::vector<PrimInfo> prims(triangles.size());
    for (unsigned i = 0; i < triangles.size(); ++i)
    {
        prims[i].index = i;
        prims[i].bounds = triangles[i].GetBounds();
        prims[i].centroid = prims[i].bounds.min * 0.5f + prims[i].bounds.max * 0.5f;
    }
    std::vector<Triangle> ordered;
    ordered.reserve(triangles.size());
    Builder b{triangles, prims, ordered, {}};
    b.pool.reserve(triangles.size() * 2);
    int root = b.Build(0, (unsigned)triangles.size());
    triangles.swap(ordered);

    nodes_.resize(b.pool.size());
    unsigned offset = 0;
    Flatten(b.pool, root, nodes_, &offset);
    assert(offset == b.pool.size());
    if (verbose)
        std::cout << "BVH created with " << nodes_.size() << " nodes for " << triangles.size() << " triangles" << std::endl;
}
} // namespace rt
