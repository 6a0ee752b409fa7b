// device_fold.h -- the interface of device_fold.hip: build_wide_bvh's SAH collapse and the adaptation's crossing counts run on the device.
#pragma once
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>
#include <stdint.h>
#include <atomic>
#include <vector>
#include "rt_types.h"
#include "wide_node.h"
#include "own_bvh.h"

namespace devfold
{
// The fold of d_nodes[nn] (device, reference layout: src/bvh.cpp:223-245) on `stream`.  On success: *d_records = a device array of *n_records WideNode (hipMalloc'ed:
// the caller frees it), entry_ref, and -- when asked for -- the records and the BVH2 node each one tests on the host.  false: the tree does not qualify
// (exactly build_wide_bvh's conditions), the metric has more than 8 directions, a device allocation failed or `cancel` was raised.
bool fold(hipStream_t stream, const rt_bvh_node* d_nodes, uint32_t nn, const rt_bvh_node& root_node, const ownbvh::Metric* metric, const double* host_weights,
    WideNode** d_records, uint32_t* n_records, uint32_t* entry_ref, std::vector<uint32_t>* roots_out, std::vector<WideNode>* records_out,
    const std::atomic<bool>* cancel = nullptr, double* seconds = nullptr);
// A binary tree over exactly the LEAVES of d_ref_nodes[nn] (device, reference layout) built on the device: PLOC with `metric` as the merge cost (ploc_kernels.h).  On success
// *d_tree = the tree in the reference's linear layout on the device (hipMalloc'ed: the caller frees it; fold() takes it as it is), *n_tree = 2 leaves - 1, and -- when asked
// for -- its host copy.  false: a leaf root, a metric of more than 8 directions, an allocation that failed, `cancel`, or the clustering did not finish within its round limit.
bool build_tree(hipStream_t stream, const rt_bvh_node* d_ref_nodes, uint32_t nn, const rt_bvh_node& root_node, const ownbvh::Metric* metric, rt_bvh_node** d_tree, uint32_t* n_tree,
    std::vector<rt_bvh_node>* tree_out, const std::atomic<bool>* cancel = nullptr, double* seconds = nullptr, uint32_t* rounds_out = nullptr,
    const float* light_dir = nullptr /* the Morton order's frame: (u, v, this direction); NULL = the world axes */, uint32_t radius = 0 /* 0 = PLOC_RADIUS */, double stretch = 1.0);
// counts[n] = probe rays whose slab test of node n passes (FoldAdapt; the host form: rtw::count_box_passes); *truncated += walks that met a subtree deeper than the stack
bool count_box_passes(hipStream_t stream, const rt_bvh_node* d_nodes, uint32_t nn, const float4* o, const float4* d, size_t n_rays, std::vector<uint32_t>& counts, uint64_t* truncated);
} // namespace devfold
