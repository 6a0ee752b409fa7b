// acceleration_structure.hpp -- the interface Integrator consumes
// (reference: src/acceleration_structure.hpp:31-38).
#pragma once
#include <vector>
#include "structures.hpp"

namespace rt
{
class AccelerationStructure
{
public:
    virtual ~AccelerationStructure() = default;
    // Builds over `triangles` and REORDERS them into leaf order (bvh.cpp:52).
    virtual void BuildCPU(std::vector<Triangle>& triangles) = 0;
    virtual std::vector<LinearBVHNode> const& GetNodes() const = 0;
};
} // namespace rt
