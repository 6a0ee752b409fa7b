// bvh.hpp -- binned-SAH BVH2 builder producing the reference's LinearBVHNode[]
// (reference: src/bvh.hpp, src/bvh.cpp:36-245).
#pragma once
#include "acceleration_structure.hpp"

namespace rt
{
class Bvh : public AccelerationStructure
{
public:
    void BuildCPU(std::vector<Triangle>& triangles) override;
    // nodes built earlier over triangles that are already in leaf order (Scene cache)
    void AdoptNodes(std::vector<LinearBVHNode> nodes) { nodes_ = std::move(nodes); }
    std::vector<LinearBVHNode> const& GetNodes() const override { return nodes_; }
    // verbose = print the two progress lines the reference prints (bvh.cpp:38,55-58)
    bool verbose = false;

private:
    std::vector<LinearBVHNode> nodes_;
};
} // namespace rt
