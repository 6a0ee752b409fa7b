// wide_bvh.h -- the host side of the backend's tree work (wide_bvh.cpp; no device code, no HIP runtime): the fold of a binary BVH into the 4-wide
// records of k_trace_w4 (build_wide_bvh: the specification of device_fold.hip's kernels, their fallback and the tests' oracle), the pair layout of the
// records, the shadow rays' metric, and the host walks of a fold adaptation (crossing counts, nearest occluders, slot order).  Split out of rt_hip.hip
// in round 6 (one 3 600-line translation unit): a change here re-links the library without touching the hot path's code object.
#pragma once
#include <stdint.h>
#include <atomic>
#include <vector>
#include <hip/hip_vector_types.h>      // float4 (the probe rays travel as the kernels' own queues)
#include "rt_types.h"
#include "wide_node.h"
#include "own_bvh.h"

namespace rtw
{
enum { RT_WIDE_TWO_LEVELS = 0, RT_WIDE_SAH = 1 };

// false: the tree does not qualify (non-finite or non-nested bounds, child order, too deep for the kernel's stack): k_trace2 is used
bool build_wide_bvh(const rt_bvh_node* nodes, uint32_t nn, int collapse, std::vector<WideNode>& out, uint32_t& entry_ref,
    std::vector<uint32_t>* roots = nullptr /* the BVH2 node each record folds (tests) */,
    const ownbvh::Metric* metric = nullptr /* what "area" means for the SAH collapse (own_bvh.h); nullptr = surface area */,
    const double* weights = nullptr /* per BVH2 node: replaces the area altogether (a MEASURED visit frequency: FoldAdapt) */,
    const std::atomic<bool>* cancel = nullptr /* set by another thread: give up (false) at the next check -- a scene uploaded again does not wait */,
    unsigned threads = 0 /* 0 = the host's (at most 32; 16 with weights); the records do not depend on it */);

// RT_CTX_OPT_WIDE_LAYOUT = 1: the records permuted into (parent, likeliest child) pairs, one pair per 128-byte line; weight[record] = the visit weight of
// the box the record tests (by OLD record index); roots (optional) is permuted along
void pair_layout(std::vector<WideNode>& wide, std::vector<uint32_t>* roots, const std::vector<double>& weight);
// ... by the area (the own trees: their metric) of the records' boxes, or by per-NODE weights (an adaptation's measured crossings)
void pair_layout_by_area(std::vector<WideNode>& wide, std::vector<uint32_t>& roots, const rt_bvh_node* nodes, uint32_t nn, const ownbvh::Metric* metric);
void pair_layout_by_node_weights(std::vector<WideNode>& wide, std::vector<uint32_t>& roots, const double* node_weights, uint32_t nn);

ownbvh::Metric shadow_metric(const rt_light* lights, uint32_t n, double iso_share);

// host threads one side of an adaptation may use beside the render loop
unsigned adapt_threads(size_t work_items, size_t per_thread);
// host walks (weights only) that met a binary tree deeper than their stack since the last call (read-and-clear) / add to it
uint64_t truncated_walks_exchange();
void truncated_walks_add(uint64_t n);

void count_box_passes(const rt_bvh_node* nodes, uint32_t nn, const float4* o, const float4* d, size_t n_rays, std::vector<uint32_t>& counts, const std::atomic<bool>& cancel);
void nearest_occluders(const std::vector<rt_bvh_node>& tree, const std::vector<float>& tri9, const std::vector<float4>& o, const std::vector<float4>& d,
    std::vector<uint32_t>& prim, const std::atomic<bool>& cancel);
uint32_t occluder_first(std::vector<WideNode>& wide, const std::vector<uint32_t>& roots, const std::vector<rt_bvh_node>& tree, const std::vector<uint32_t>& prim);
} // namespace rtw
