// relayout_kernels.h -- the reference's scene arrays -> traversal / shading records, on the device.
#pragma once
#include "kernels_common.h"

// ---------------------------------------------------------------------------
// Scene re-layout on the device (rt_scene_upload): the reference's arrays are copied to HBM
// as they are and three streaming kernels write the traversal / shading layouts.
// Error codes (first one wins) are decoded by the host.
// ---------------------------------------------------------------------------
enum { RL_OK = 0, RL_CHILD_RANGE = 1, RL_LEAF_RANGE = 2, RL_AXIS = 3, RL_MATERIAL = 4 };

// one thread per LinearBVHNode: leaves mark their last triangle
__global__ void k_relayout_mark_leaves(const rt_bvh_node* __restrict__ nodes, uint32_t nn, uint32_t nt,
    uint8_t* __restrict__ last_in_leaf, int* __restrict__ err)
{
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nn) return;
    uint32_t n = nodes[i].num_primitives_axis >> 16;
    if (n == 0) return;
    uint32_t first = nodes[i].offset;
    if ((unsigned long long)first + n > nt) { atomicCAS(err, RL_OK, RL_LEAF_RANGE); return; }
    last_in_leaf[first + n - 1u] = 1;
}

// one thread per LinearBVHNode: interior nodes write their child-pair record
__global__ void k_relayout_nodes(const rt_bvh_node* __restrict__ nodes, uint32_t nn,
    const uint32_t* __restrict__ interior_index, float4* __restrict__ out_nodes, int* __restrict__ err)
{
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nn) return;
    const rt_bvh_node nd = nodes[i];
    if ((nd.num_primitives_axis >> 16) != 0) return;
    if (interior_index[i] == RT_EMPTY_REF) return;           // not reachable from the root
    uint32_t c0 = i + 1u, c1 = nd.offset;                    // first child follows, second child at offset
    if (c0 >= nn || c1 >= nn) { atomicCAS(err, RL_OK, RL_CHILD_RANGE); return; }
    uint32_t axis = nd.num_primitives_axis & 0xFFFFu;
    if (axis > 2u) { atomicCAS(err, RL_OK, RL_AXIS); return; }
    const rt_bvh_node a = nodes[c0], b = nodes[c1];
    uint32_t r0 = (a.num_primitives_axis >> 16) ? (RT_LEAF_BIT | a.offset) : interior_index[c0];
    uint32_t r1 = (b.num_primitives_axis >> 16) ? (RT_LEAF_BIT | b.offset) : interior_index[c1];
    float4* out = out_nodes + (size_t)interior_index[i] * 4;
    // pair layout of trace_kernels.h (RT_NODE_C0 / RT_NODE_C1): x,y corners packed, z planes together
    out[0] = make_float4(a.bounds_min.x, a.bounds_min.y, a.bounds_max.x, a.bounds_max.y);
    out[1] = make_float4(b.bounds_min.x, b.bounds_min.y, b.bounds_max.x, b.bounds_max.y);
    out[2] = make_float4(a.bounds_min.z, a.bounds_max.z, b.bounds_min.z, b.bounds_max.z);
    out[3] = make_float4(__uint_as_float(r0), __uint_as_float(r1), __uint_as_float(axis), 0.0f);
}

// one thread per triangle: 64-byte trace record (p1, e1, e2) and 128-byte shading record
__global__ void k_relayout_triangles(const rt_triangle* __restrict__ tris, uint32_t nt, uint32_t num_materials,
    const uint8_t* __restrict__ last_in_leaf, float4* __restrict__ trt, float4* __restrict__ tsh, int* __restrict__ err)
{
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nt) return;
    const rt_triangle t = tris[i];
    const rt_float3 p1 = t.v1.position, p2 = t.v2.position, p3 = t.v3.position;
    float4* r = trt + (size_t)i * 4;
    r[0] = make_float4(p1.x, p1.y, p1.z, last_in_leaf[i] ? 1.0f : 0.0f);
    r[1] = make_float4(p2.x - p1.x, p2.y - p1.y, p2.z - p1.z, 0.0f);      // e1, trace_bvh.cl:30
    r[2] = make_float4(p3.x - p1.x, p3.y - p1.y, p3.z - p1.z, 0.0f);      // e2, trace_bvh.cl:31
    r[3] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);                           // pad to one 64-byte line
    float4* q = tsh + (size_t)i * 8;
    q[0] = make_float4(p1.x, p1.y, p1.z, t.v1.texcoord.x);
    q[1] = make_float4(p2.x, p2.y, p2.z, t.v1.texcoord.y);
    q[2] = make_float4(p3.x, p3.y, p3.z, t.v2.texcoord.x);
    q[3] = make_float4(t.v1.normal.x, t.v1.normal.y, t.v1.normal.z, t.v2.texcoord.y);
    q[4] = make_float4(t.v2.normal.x, t.v2.normal.y, t.v2.normal.z, t.v3.texcoord.x);
    q[5] = make_float4(t.v3.normal.x, t.v3.normal.y, t.v3.normal.z, t.v3.texcoord.y);
    if (t.mtl_index >= num_materials) atomicCAS(err, RL_OK, RL_MATERIAL);
    q[6] = make_float4(__uint_as_float(t.mtl_index), 0.0f, 0.0f, 0.0f);
    q[7] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// one thread per LinearBVHNode, after k_relayout_triangles: a leaf writes its exact bounds into the spare
// floats of its triangles' trace records (k_trace_w4 re-tests a leaf's box with them when it reaches it):
//   r[1].w = min.x, r[2].w = min.y, r[3] = (min.z, max.x, max.y, max.z)
__global__ void k_relayout_leaf_bounds(const rt_bvh_node* __restrict__ nodes, uint32_t nn, uint32_t nt, float4* __restrict__ trt)
{
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nn) return;
    const rt_bvh_node nd = nodes[i];
    uint32_t n = nd.num_primitives_axis >> 16;
    if (n == 0) return;
    uint32_t first = nd.offset;
    if ((unsigned long long)first + n > nt) return;          // reported by k_relayout_mark_leaves
    for (uint32_t k = 0; k < n; ++k)
    {
        float* r = reinterpret_cast<float*>(trt + (size_t)(first + k) * 4);
        r[7] = nd.bounds_min.x;
        r[11] = nd.bounds_min.y;
        r[12] = nd.bounds_min.z; r[13] = nd.bounds_max.x; r[14] = nd.bounds_max.y; r[15] = nd.bounds_max.z;
    }
}
