// ploc_kernels.h -- a binary tree over the reference's LEAVES built on the device (round 6; SURVEY 8f-1: "a HIP LBVH / PLOC").
//
// own_bvh.h builds the shadow rays' own tree with a full-sweep SAH on host threads (0.39 s for the headline scene's 2.45 M leaves, 1.7 s for 8.7 M:
// the longest stage of rt_scene_upload once the fold runs on the device).  Any binary tree over exactly the reference's leaves gives an any-hit query the
// reference's verdict (own_bvh.h: interior culling is conservative, leaves are decided by the reference's expression on their exact boxes), so the
// builder is free -- here it is PLOC (Meister & Bittner 2017: parallel locally-ordered clustering), bottom-up, with the metric the tree is built for
// (projected area along the directional lights + an isotropic share: FoldMetric) as the merge cost:
//
//   k_ploc_flag_leaves + scan   the reference's leaves in node order
//   k_ploc_keys + radix sort    63-bit Morton codes of the leaf boxes' centres (hipcub::DeviceRadixSort, stable: equal codes keep node order)
//   per round:  k_ploc_nearest  every cluster's cheapest partner within +- RADIUS positions (cost = metric of the union box; ties: the lower position)
//               k_ploc_decide   mutual choices merge (the lower position becomes the new cluster), + two scans: positions and node ids, both in position order
//               k_ploc_apply    new interior nodes, the compacted cluster list
//   k_ploc_sizes                subtree sizes, bottom-up (the second arrival at a node computes it)
//   k_ploc_emit                 the reference's linear layout (src/bvh.cpp:223-245): a node's index = the nodes before it in pre-order, found by climbing
//                               to the root; first child at i + 1, second at `offset`; leaves are the reference's records (axis bits cleared, like own_bvh.h)
// Every choice is a function of positions, never of thread arrival, so the tree is the same every run.  fold_kernels.h folds it; rt_scene_upload's choice by
// proxy rays measures it against the reference's topology like any other candidate; the adaptation rotates it (tree_rotate.h) like the host-built one.
#pragma once
#include "fold_kernels.h"

#define PLOC_RADIUS 32

struct PlocNode { float mn[3]; uint32_t left; float mx[3]; uint32_t right; };     // interior: children (pool ids); leaf: left = reference node | 0x80000000, right = unused

RT_DEV double ploc_cost(const FoldMetric& m, const float* amn, const float* amx, const float* bmn, const float* bmx)
{
    const double dx = (double)fmaxf(amx[0], bmx[0]) - (double)fminf(amn[0], bmn[0]);
    const double dy = (double)fmaxf(amx[1], bmx[1]) - (double)fminf(amn[1], bmn[1]);
    const double dz = (double)fmaxf(amx[2], bmx[2]) - (double)fminf(amn[2], bmn[2]);
    if (m.n_dirs < 0) return dx * dy + dy * dz + dz * dx;
    double c = m.iso * 0.5 * (dx * dy + dy * dz + dz * dx);
    for (int i = 0; i < m.n_dirs; ++i) c += m.dirs[i][0] * dy * dz + m.dirs[i][1] * dz * dx + m.dirs[i][2] * dx * dy;
    return c;
}

__global__ __launch_bounds__(256) void k_ploc_flag_leaves(const rt_bvh_node* __restrict__ nodes, uint32_t nn, uint32_t* __restrict__ flags)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < nn) flags[i] = (nodes[i].num_primitives_axis >> 16) != 0u ? 1u : 0u;
}

RT_DEV unsigned long long ploc_spread21(uint32_t v)       // 21 bits -> every third bit
{
    unsigned long long x = v & 0x1FFFFFull;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

// leaf_nodes[k] = the k-th leaf of the reference's array; key = Morton code of its box's centre in the root's box; the leaf becomes pool node k' after the sort
// The order the clustering starts from: Morton codes of the leaf boxes' centres in a FRAME (rows of `frame`: three directions, each with the interval [lo, hi] the
// scene's corners span along it).  The world axes for an isotropic metric; for shadow rays towards a directional light a frame with the light's direction as its
// third axis, that axis' cells `stretch` times as long as the others' -- boxes that are long along the light are what that metric prices cheaply, and leaves that
// follow each other along the light should be neighbours in the order (measured: DESIGN.md section 7).
struct PlocFrame { double axis[3][3]; double lo[3], hi[3]; };
__global__ __launch_bounds__(256) void k_ploc_keys(const rt_bvh_node* __restrict__ nodes, const uint32_t* __restrict__ leaf_nodes, uint32_t n_leaves, PlocFrame frame,
    unsigned long long* __restrict__ keys, uint32_t* __restrict__ values)
{
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= n_leaves) return;
    const rt_bvh_node b = nodes[leaf_nodes[k]];
    const double w[3] = {0.5 * ((double)b.bounds_min.x + (double)b.bounds_max.x), 0.5 * ((double)b.bounds_min.y + (double)b.bounds_max.y), 0.5 * ((double)b.bounds_min.z + (double)b.bounds_max.z)};
    uint32_t q[3];
    for (int a = 0; a < 3; ++a)
    {
        const double c = frame.axis[a][0] * w[0] + frame.axis[a][1] * w[1] + frame.axis[a][2] * w[2];
        const double e = frame.hi[a] - frame.lo[a];
        double t = e > 0.0 ? (c - frame.lo[a]) / e : 0.0;
        t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
        const double s = t * 2097151.0;
        q[a] = (uint32_t)s;
    }
    keys[k] = ploc_spread21(q[0]) | ploc_spread21(q[1]) << 1 | ploc_spread21(q[2]) << 2;
    values[k] = k;
}

// the sorted leaves become pool nodes 0 .. n_leaves - 1 and the first cluster list
__global__ __launch_bounds__(256) void k_ploc_init(const rt_bvh_node* __restrict__ nodes, const uint32_t* __restrict__ leaf_nodes, const uint32_t* __restrict__ sorted_values,
    uint32_t n_leaves, PlocNode* __restrict__ pool, uint32_t* __restrict__ parent, uint32_t* __restrict__ cluster)
{
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= n_leaves) return;
    const uint32_t ref = leaf_nodes[sorted_values[k]];
    const rt_bvh_node b = nodes[ref];
    PlocNode p;
    p.mn[0] = b.bounds_min.x; p.mn[1] = b.bounds_min.y; p.mn[2] = b.bounds_min.z;
    p.mx[0] = b.bounds_max.x; p.mx[1] = b.bounds_max.y; p.mx[2] = b.bounds_max.z;
    p.left = ref | 0x80000000u; p.right = 0u;
    pool[k] = p;
    parent[k] = RT_EMPTY_REF;
    cluster[k] = k;
}

__global__ __launch_bounds__(256) void k_ploc_nearest(const PlocNode* __restrict__ pool, const uint32_t* __restrict__ cluster, uint32_t n, FoldMetric metric, uint32_t radius, uint32_t* __restrict__ nearest)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const PlocNode a = pool[cluster[i]];
    const uint32_t lo = i > radius ? i - radius : 0u, hi = i + radius < n - 1u ? i + radius : n - 1u;
    double best = 0.0; uint32_t at = RT_EMPTY_REF;
    for (uint32_t j = lo; j <= hi; ++j)
    {
        if (j == i) continue;
        const PlocNode b = pool[cluster[j]];
        const double c = ploc_cost(metric, a.mn, a.mx, b.mn, b.mx);
        if (at == RT_EMPTY_REF || c < best) { best = c; at = j; }         // (ascending j: ties go to the lower position)
    }
    nearest[i] = at;
}

// keep[i] = the cluster at position i survives the round (it is not the upper half of a mutual pair); merge[i] = it is the lower half: a new node
__global__ __launch_bounds__(256) void k_ploc_decide(const uint32_t* __restrict__ nearest, uint32_t n, uint32_t* __restrict__ keep, uint32_t* __restrict__ merge)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = nearest[i];
    const bool mutual = j != RT_EMPTY_REF && j < n && nearest[j] == i;
    keep[i] = mutual && j < i ? 0u : 1u;
    merge[i] = mutual && i < j ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_ploc_apply(PlocNode* __restrict__ pool, uint32_t* __restrict__ parent, const uint32_t* __restrict__ cluster, const uint32_t* __restrict__ nearest,
    const uint32_t* __restrict__ keep, const uint32_t* __restrict__ merge, const uint32_t* __restrict__ keep_at, const uint32_t* __restrict__ merge_at, uint32_t n, uint32_t first_new_id,
    uint32_t* __restrict__ cluster_out)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n || !keep[i]) return;
    uint32_t id = cluster[i];
    if (merge[i])
    {
        const uint32_t l = id, r = cluster[nearest[i]];
        const PlocNode a = pool[l], b = pool[r];
        id = first_new_id + merge_at[i];
        PlocNode p;
        for (int k = 0; k < 3; ++k) { p.mn[k] = fminf(a.mn[k], b.mn[k]); p.mx[k] = fmaxf(a.mx[k], b.mx[k]); }
        p.left = l; p.right = r;
        pool[id] = p;
        parent[id] = RT_EMPTY_REF;
        parent[l] = id; parent[r] = id;
    }
    cluster_out[keep_at[i]] = id;
}

// subtree sizes in nodes, bottom-up (fold_kernels.h's pattern: the second arrival computes)
__global__ __launch_bounds__(256) void k_ploc_sizes(const PlocNode* __restrict__ pool, const uint32_t* __restrict__ parent, uint32_t n_leaves, uint32_t n_nodes, uint32_t* __restrict__ arrived,
    uint32_t* __restrict__ size, int* __restrict__ error)
{
    const uint32_t leaf = blockIdx.x * 256u + threadIdx.x;
    if (leaf >= n_leaves) return;
    size[leaf] = 1u;
    uint32_t n = parent[leaf], guard = 0;
    while (n != RT_EMPTY_REF && n < n_nodes)
    {
        if (++guard > 100000u) { *error = FOLD_NOT_A_TREE; return; }
        __threadfence();
        if (atomicAdd(&arrived[n], 1u) == 0u) return;
        __threadfence();
        const uint32_t l = pool[n].left, r = pool[n].right;
        size[n] = 1u + __hip_atomic_load(&size[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + __hip_atomic_load(&size[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        n = parent[n];
    }
}

// pool node -> the reference's linear layout.  index(n) = nodes before n in pre-order = sum over the climb to the root of (1 + the left sibling's subtree when n hangs on the right)
__global__ __launch_bounds__(256) void k_ploc_emit(const rt_bvh_node* __restrict__ ref_nodes, const PlocNode* __restrict__ pool, const uint32_t* __restrict__ parent, const uint32_t* __restrict__ size,
    uint32_t n_nodes, uint32_t root, rt_bvh_node* __restrict__ out, uint32_t* __restrict__ index_of, int* __restrict__ error)
{
    const uint32_t n = blockIdx.x * 256u + threadIdx.x;
    if (n >= n_nodes) return;
    uint32_t idx = 0, a = n, guard = 0;
    while (a != root)
    {
        const uint32_t p = parent[a];
        if (p == RT_EMPTY_REF || p >= n_nodes || ++guard > 100000u) { *error = FOLD_NOT_A_TREE; return; }
        idx += 1u + (pool[p].right == a ? size[pool[p].left] : 0u);
        a = p;
    }
    index_of[n] = idx;
    const PlocNode me = pool[n];
    rt_bvh_node o;
    if (me.left & 0x80000000u)
    {
        o = ref_nodes[me.left & 0x7FFFFFFFu];                              // the reference's leaf record: exact box, first triangle, count
        o.num_primitives_axis &= 0xFFFF0000u;
    }
    else
    {
        const PlocNode l = pool[me.left], r = pool[me.right];
        o.bounds_min.x = me.mn[0]; o.bounds_min.y = me.mn[1]; o.bounds_min.z = me.mn[2]; o.bounds_min.w = 0.0f;
        o.bounds_max.x = me.mx[0]; o.bounds_max.y = me.mx[1]; o.bounds_max.z = me.mx[2]; o.bounds_max.w = 0.0f;
        o.offset = idx + 1u + size[me.left];                               // the second child follows the first child's subtree
        // the axis along which the children's centres lie furthest apart (the reference's field; an any-hit walk does not read it)
        float best = -1.0f; uint32_t axis = 0;
        for (uint32_t k = 0; k < 3u; ++k)
        {
            const float d = fabsf((l.mn[k] + l.mx[k]) - (r.mn[k] + r.mx[k]));
            if (d > best) { best = d; axis = k; }
        }
        o.num_primitives_axis = axis;
        o.padding[0] = 0u; o.padding[1] = 0u;
    }
    out[idx] = o;
}
