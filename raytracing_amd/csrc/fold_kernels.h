// fold_kernels.h -- build_wide_bvh on the DEVICE (round 6; SURVEY 8f-1: "or at least a parallel collapse to BVH4/8").
//
// The fold of a binary BVH in the reference's linear layout (LinearBVHNode[], src/bvh.cpp:223-245: first child at n + 1, second at
// `offset`) into the 64-byte 4-wide records of k_trace_w4 -- the SAH-optimal frontier per record (a dynamic programme over node x slots,
// Ylitie / Karras / Laine 2017, section 4.1), the placement of the slots for the kernel's exchange network, the 8-bit boxes on a per-record
// power-of-two grid rounded outward.  wide_bvh.cpp holds the same algorithm for the host (build_wide_bvh: the specification, the fallback
// and the tests' oracle -- tests/test_gpu_device_fold.py compares the two record for record); here it is five kernels:
//
//   k_fold_prepare   per node: validation (finite, nested bounds, children behind their parent), parent links
//   k_fold_dp        bottom-up: one thread per leaf climbs; the SECOND arrival at a node computes it (T[n], F[n][2..4], the split table,
//                    the `open` bits) from its children's values -- the sweep the host makes backwards over the array
//   k_fold_roots     top-down, one launch per level of the 4-wide tree: the interior slots of every record root become record roots
//   k_fold_scan_*    record index = rank of the node among the record roots in node order -- which IS the host's depth-first order
//                    (slot 0's subtree first), because the reference's layout is a pre-order
//   k_fold_emit      per record: slots, visit orders per direction octant, placement + exchange bits, frame, quantised boxes, refs
//
// Exactness does not depend on any of it (every fold of the same binary tree tests the same leaves in the same order: DESIGN.md section 2);
// the arithmetic is the host's all the same (binary64, -ffp-contract=off), so that the two builds can be compared bit for bit.
// k_count_box_passes: the adaptation's ray-box crossing counts (FoldAdapt: count_box_passes on the host) as one thread per probe ray.
#pragma once
#include "kernels_common.h"
#include "wide_node.h"

// what "area" means to the collapse: the host's ownbvh::Metric (iso * half the surface area + projected areas along <= 8 directions);
// n_dirs < 0: the plain surface area dx dy + dy dz + dz dx (build_wide_bvh without a metric)
struct FoldMetric { double iso; double dirs[8][3]; int n_dirs; };

enum { FOLD_OK = 0, FOLD_NOT_FINITE = 1, FOLD_NOT_NESTED = 2, FOLD_BAD_CHILD = 3, FOLD_TOO_DEEP = 4, FOLD_NO_ARRANGEMENT = 5, FOLD_FRAME = 6, FOLD_NOT_A_TREE = 7 };

struct FoldState
{
    const rt_bvh_node* nodes; uint32_t nn;
    uint32_t* parent;             // [nn]
    uint32_t* arrived;            // [nn] bottom-up arrival counters
    double* T;                    // [nn]
    double* F;                    // [nn][3]: k = 2, 3, 4
    uint8_t* split;               // [nn][3]: slots given to the first child for k = 2, 3, 4
    uint8_t* open;                // [nn]: bit k set = with k slots to spend the node is folded rather than kept as one slot
    uint32_t* is_root;            // [nn] 1 = a record is rooted here; after the scan: wide_of (exclusive prefix sum)
    uint8_t* depth;               // [nn] depth of the record rooted here (root = 1)
    const double* weights;        // per node, or NULL
    FoldMetric metric;
    int* error;
};

RT_DEV bool fold_leaf(const rt_bvh_node* nodes, uint32_t i) { return (nodes[i].num_primitives_axis >> 16) != 0u; }

RT_DEV double fold_weight(const FoldState& s, uint32_t n)
{
    if (s.weights) return s.weights[n];
    const rt_bvh_node& b = s.nodes[n];
    const double dx = (double)b.bounds_max.x - b.bounds_min.x, dy = (double)b.bounds_max.y - b.bounds_min.y, dz = (double)b.bounds_max.z - b.bounds_min.z;
    if (s.metric.n_dirs < 0) return dx * dy + dy * dz + dz * dx;
    double m = s.metric.iso * 0.5 * (dx * dy + dy * dz + dz * dx);
    for (int i = 0; i < s.metric.n_dirs; ++i) m += s.metric.dirs[i][0] * dy * dz + s.metric.dirs[i][1] * dz * dx + s.metric.dirs[i][2] * dx * dy;
    return m;
}

__global__ __launch_bounds__(256) void k_fold_prepare(FoldState s)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= s.nn) return;
    const rt_bvh_node n = s.nodes[i];
    s.arrived[i] = 0u;
    s.is_root[i] = 0u;
    s.depth[i] = 0;
    if (i == 0u) s.parent[0] = RT_EMPTY_REF;
    auto finite3 = [](const rt_float3& v) { return __builtin_isfinite(v.x) && __builtin_isfinite(v.y) && __builtin_isfinite(v.z); };
    if (!finite3(n.bounds_min) || !finite3(n.bounds_max)) { *s.error = FOLD_NOT_FINITE; return; }
    if ((n.num_primitives_axis >> 16) != 0u) return;
    if (i + 1u >= s.nn || n.offset >= s.nn || n.offset <= i + 1u || (n.num_primitives_axis & 0xFFFFu) > 2u) { *s.error = FOLD_BAD_CHILD; return; }
    const uint32_t kids[2] = {i + 1u, n.offset};
    for (int c = 0; c < 2; ++c)
    {
        const rt_bvh_node k = s.nodes[kids[c]];
        if (k.bounds_min.x < n.bounds_min.x || k.bounds_min.y < n.bounds_min.y || k.bounds_min.z < n.bounds_min.z ||
            k.bounds_max.x > n.bounds_max.x || k.bounds_max.y > n.bounds_max.y || k.bounds_max.z > n.bounds_max.z) { *s.error = FOLD_NOT_NESTED; return; }
        s.parent[kids[c]] = i;
    }
}

// values another thread (another CU) has written and published with a fence: read past this CU's L1
RT_DEV double fold_read(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(256) void k_fold_dp(FoldState s)
{
    const uint32_t leaf = blockIdx.x * 256u + threadIdx.x;
    if (leaf >= s.nn || !fold_leaf(s.nodes, leaf)) return;
    uint32_t n = s.parent[leaf];
    uint32_t guard = 0;
    while (n != RT_EMPTY_REF && n < s.nn)
    {
        if (++guard > 4096u) { *s.error = FOLD_NOT_A_TREE; return; }
        __threadfence();                                           // this subtree's values, before the arrival that may hand them on
        if (atomicAdd(&s.arrived[n], 1u) == 0u) return;            // the first of the two: the other one goes on
        __threadfence();
        const uint32_t l = n + 1u, r = s.nodes[n].offset;
        const bool ll = fold_leaf(s.nodes, l), rl = fold_leaf(s.nodes, r);
        double Tl = 0.0, Tr = 0.0, Fl[3] = {0.0, 0.0, 0.0}, Fr[3] = {0.0, 0.0, 0.0};
        if (!ll) { Tl = fold_read(&s.T[l]); for (int k = 0; k < 3; ++k) Fl[k] = fold_read(&s.F[(size_t)l * 3u + k]); }
        if (!rl) { Tr = fold_read(&s.T[r]); for (int k = 0; k < 3; ++k) Fr[k] = fold_read(&s.F[(size_t)r * 3u + k]); }
        // G(c, i): cost of c's subtree given i slots -- a leaf costs nothing, one slot is a record of its own, more may be folded or not
        auto G = [](bool leafc, double T, const double (&F)[3], uint32_t i) { return leafc ? 0.0 : (i >= 2u ? (T < F[i - 2u] ? T : F[i - 2u]) : T); };
        double Fn[3];
        for (uint32_t k = 2; k <= 4; ++k)
        {
            double best = 0.0; uint32_t at = 0;
            for (uint32_t i = 1; i < k; ++i)
            {
                const double c = G(ll, Tl, Fl, i) + G(rl, Tr, Fr, k - i);
                if (at == 0u || c < best) { best = c; at = i; }
            }
            Fn[k - 2u] = best;
            s.split[(size_t)n * 3u + (k - 2u)] = (uint8_t)at;
        }
        const double Tn = fold_weight(s, n) + Fn[2];
        uint8_t open = 0;
        for (uint32_t i = 2; i <= 4; ++i) if (Fn[i - 2u] < Tn) open |= (uint8_t)(1u << i);
        s.open[n] = open;
        for (int k = 0; k < 3; ++k) s.F[(size_t)n * 3u + k] = Fn[k];
        s.T[n] = Tn;
        n = s.parent[n];
    }
}

// The record rooted at BVH2 node `root`: its slots in the binary tree's depth-first order and, per direction octant, their positions in the
// reference's visit order (the second child first where the ray is negative along the folded node's split axis, trace_bvh.cl:181-190).
struct FoldRecord { uint32_t slot[4]; uint32_t n_slots; uint8_t visit[8][4]; };

// one walk of the folded piece below `root`: octant < 0 numbers the slots (depth-first, first child first), octant >= 0 lists them in that octant's visit order
RT_DEV void fold_walk(const rt_bvh_node* nodes, const uint8_t* split, const uint8_t* open, uint32_t root, int octant, FoldRecord& f)
{
    struct Item { uint32_t node; uint32_t k; };                    // k = slots to spend below this node
    Item stack[8];
    int sp = 0;
    uint32_t at = 0;
    stack[sp++] = {root, 4u};
    bool first = true;
    while (sp > 0)
    {
        const Item it = stack[--sp];
        // the root is folded by definition; a node below it iff it is interior, has two or more slots to spend and the DP opened it for that many
        const bool folded = first || (!fold_leaf(nodes, it.node) && it.k >= 2u && ((open[it.node] >> it.k) & 1u));
        first = false;
        if (!folded)
        {
            if (octant < 0) { if (f.n_slots < 4u) f.slot[f.n_slots++] = it.node; }
            else
            {
                uint32_t idx = 0;
                for (uint32_t j = 0; j < f.n_slots; ++j) if (f.slot[j] == it.node) idx = j;
                if (at < 4u) f.visit[octant][at++] = (uint8_t)idx;
            }
            continue;
        }
        const uint32_t give0 = split[(size_t)it.node * 3u + (it.k - 2u)];
        const Item c0 = {it.node + 1u, give0}, c1 = {nodes[it.node].offset, it.k - give0};
        const uint32_t axis = nodes[it.node].num_primitives_axis & 0xFFFFu;
        const bool second_first = octant >= 0 && (((uint32_t)octant >> axis) & 1u);     // trace_bvh.cl:181-190
        if (sp + 2 > 8) return;
        if (second_first) { stack[sp++] = c0; stack[sp++] = c1; }      // popped: c1, then c0
        else { stack[sp++] = c1; stack[sp++] = c0; }
    }
}

RT_DEV void fold_of(const rt_bvh_node* nodes, const uint8_t* split, const uint8_t* open, uint32_t root, FoldRecord& f)
{
    for (int k = 0; k < 4; ++k) f.slot[k] = RT_EMPTY_REF;
    f.n_slots = 0;
    for (int o = 0; o < 8; ++o) for (int k = 0; k < 4; ++k) f.visit[o][k] = 0;
    fold_walk(nodes, split, open, root, -1, f);
    for (int o = 0; o < 8; ++o) fold_walk(nodes, split, open, root, o, f);
}

__global__ __launch_bounds__(256) void k_fold_roots(FoldState s, const uint32_t* __restrict__ frontier, uint32_t n_frontier, uint32_t* __restrict__ next, uint32_t* __restrict__ n_next,
    uint32_t level)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_frontier) return;
    const uint32_t n = frontier[i];
    if (atomicExch(&s.is_root[n], 1u) != 0u) { *s.error = FOLD_NOT_A_TREE; return; }     // reached twice: several parents share a child
    s.depth[n] = (uint8_t)level;
    FoldRecord f;
    fold_of(s.nodes, s.split, s.open, n, f);
    for (uint32_t k = 0; k < f.n_slots; ++k)
        if (!fold_leaf(s.nodes, f.slot[k])) next[atomicAdd(n_next, 1u)] = f.slot[k];
}

// ---- exclusive prefix sum of is_root (in place: is_root[n] becomes the record index of n), three small kernels ----
#define FOLD_SCAN_BLOCK 1024u
__global__ __launch_bounds__(256) void k_fold_scan_sums(const uint32_t* __restrict__ flags, uint32_t n, uint32_t* __restrict__ block_sums)
{
    __shared__ uint32_t part[256];
    const uint32_t base = blockIdx.x * FOLD_SCAN_BLOCK;
    uint32_t sum = 0;
    for (uint32_t k = 0; k < 4u; ++k) { const uint32_t i = base + k * 256u + threadIdx.x; if (i < n) sum += flags[i]; }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t st = 128u; st > 0u; st >>= 1) { if (threadIdx.x < st) part[threadIdx.x] += part[threadIdx.x + st]; __syncthreads(); }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = part[0];
}
__global__ __launch_bounds__(1024) void k_fold_scan_blocks(uint32_t* __restrict__ block_sums, uint32_t n_blocks, uint32_t* __restrict__ total)
{
    // one block walks the block sums in tiles of 1024 (a few thousand of them at most)
    __shared__ uint32_t tile[1024];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_blocks; base += 1024u)
    {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n_blocks ? block_sums[i] : 0u;
        tile[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t d = 1; d < 1024u; d <<= 1)
        {
            const uint32_t t = threadIdx.x >= d ? tile[threadIdx.x - d] : 0u;
            __syncthreads();
            tile[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n_blocks) block_sums[i] = carry + tile[threadIdx.x] - v;     // exclusive
        __syncthreads();
        if (threadIdx.x == 1023u) carry += tile[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}
__global__ __launch_bounds__(256) void k_fold_scan_apply(uint32_t* __restrict__ flags, uint32_t n, const uint32_t* __restrict__ block_offsets, uint32_t* __restrict__ roots)
{
    // each thread owns 4 consecutive elements of its block's 1024
    __shared__ uint32_t part[256];
    const uint32_t base = blockIdx.x * FOLD_SCAN_BLOCK + threadIdx.x * 4u;
    uint32_t v[4], sum = 0;
    for (uint32_t k = 0; k < 4u; ++k) { v[k] = base + k < n ? flags[base + k] : 0u; sum += v[k]; }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 256u; d <<= 1)
    {
        const uint32_t t = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = block_offsets[blockIdx.x] + part[threadIdx.x] - sum;
    for (uint32_t k = 0; k < 4u; ++k)
    {
        if (base + k >= n) break;
        if (v[k]) { roots[run] = base + k; flags[base + k] = run; run += 1u; }
        else flags[base + k] = RT_EMPTY_REF;
    }
}

// Where the slots of a record are stored: the kernel brings them into visit order with four conditional exchanges -- (0,1), (2,3), (0,2), (1,3),
// one table bit each per direction octant -- and for every shape a record can fold there is a placement for which that network realises all the
// orders the shape can ask for (tests/test_wide_bvh.py).  The host tries the 24 placements in lexicographic order, depth-first order first,
// once per distinct key; here every record does (the same deterministic search, so the same answer).
RT_DEV bool fold_arrange(const FoldRecord& f, uint8_t (&place_out)[4], uint32_t& order_out)
{
    uint8_t place[4] = {0, 1, 2, 3};
    for (int perm = 0; perm < 24; ++perm)
    {
        uint8_t node_at[4] = {255, 255, 255, 255};
        for (uint32_t j = 0; j < f.n_slots; ++j) node_at[place[j]] = (uint8_t)j;
        uint32_t order = 0;
        bool all = true;
        for (uint32_t o = 0; o < 8u && all; ++o)
        {
            bool found = false;
            for (uint32_t bits = 0; bits < 16u && !found; ++bits)
            {
                uint8_t pos[4] = {0, 1, 2, 3};
                if (bits & 1u) { const uint8_t t = pos[0]; pos[0] = pos[1]; pos[1] = t; }
                if (bits & 2u) { const uint8_t t = pos[2]; pos[2] = pos[3]; pos[3] = t; }
                if (bits & 4u) { const uint8_t t = pos[0]; pos[0] = pos[2]; pos[2] = t; }
                if (bits & 8u) { const uint8_t t = pos[1]; pos[1] = pos[3]; pos[3] = t; }
                uint32_t at = 0;
                bool same = true;
                for (int k = 0; k < 4 && same; ++k)
                    if (node_at[pos[k]] != 255) same = node_at[pos[k]] == f.visit[o][at++];
                if (same) { order |= bits << (4u * o); found = true; }
            }
            all = found;
        }
        if (all) { for (int k = 0; k < 4; ++k) place_out[k] = place[k]; order_out = order; return true; }
        // std::next_permutation
        int i = 2;
        while (i >= 0 && place[i] >= place[i + 1]) --i;
        if (i < 0) break;
        int j = 3;
        while (place[j] <= place[i]) --j;
        { const uint8_t t = place[i]; place[i] = place[j]; place[j] = t; }
        for (int a = i + 1, b = 3; a < b; ++a, --b) { const uint8_t t = place[a]; place[a] = place[b]; place[b] = t; }
    }
    return false;
}

__global__ __launch_bounds__(64) void k_fold_emit(FoldState s, const uint32_t* __restrict__ roots, uint32_t n_records, WideNode* __restrict__ out)
{
    const uint32_t w = blockIdx.x * 64u + threadIdx.x;
    if (w >= n_records) return;
    const rt_bvh_node* nodes = s.nodes;
    const uint32_t n = roots[w];
    FoldRecord f;
    fold_of(nodes, s.split, s.open, n, f);
    uint8_t place[4]; uint32_t order_bits = 0;
    if (!fold_arrange(f, place, order_bits)) { *s.error = FOLD_NO_ARRANGEMENT; return; }
    uint32_t slot[4] = {RT_EMPTY_REF, RT_EMPTY_REF, RT_EMPTY_REF, RT_EMPTY_REF};
    for (uint32_t j = 0; j < f.n_slots; ++j) slot[place[j]] = f.slot[j];
    WideNode r;
    r.pad = 0; r.lo[0] = r.lo[1] = r.lo[2] = 0; r.hi[0] = r.hi[1] = r.hi[2] = 0;
    const float nmin[3] = {nodes[n].bounds_min.x, nodes[n].bounds_min.y, nodes[n].bounds_min.z};
    const float nmax[3] = {nodes[n].bounds_max.x, nodes[n].bounds_max.y, nodes[n].bounds_max.z};
    float origin[3]; int exps[3];
    for (int a = 0; a < 3; ++a)
    {
        // cell = 2^e: 254 cells span the node (one spare for the floor of the origin); the grid stays exactly representable
        // (|origin| / cell < 2^23 leaves room for + 255 below 2^24): build_wide_bvh's rule, the exponent found without a logarithm
        const double extent = (double)nmax[a] - (double)nmin[a];
        const double amax = fmax(fabs((double)nmin[a]), fabs((double)nmax[a]));
        int e = -126;
        if (extent > 0.0)
        {
            int ex = 0;
            const double m = frexp(extent / 254.0, &ex);           // extent / 254 = m * 2^ex, m in [0.5, 1): ceil(log2) = ex, or ex - 1 for a power of two
            const int c = m == 0.5 ? ex - 1 : ex;
            e = c > e ? c : e;
        }
        while (ldexp(254.0, e) < extent) ++e;
        while (amax > 0.0 && amax / ldexp(1.0, e) >= 8388608.0 - 256.0) ++e;
        if (e > 20 || amax >= 268435456.0) { *s.error = FOLD_FRAME; return; }
        const double cell = ldexp(1.0, e);
        const double o = floor((double)nmin[a] / cell) * cell;
        origin[a] = (float)o;
        if ((double)origin[a] != o) { *s.error = FOLD_FRAME; return; }
        exps[a] = e;
    }
    r.ox = origin[0]; r.oy = origin[1]; r.oz = origin[2];
    r.meta = (uint32_t)(exps[0] + 127) | (uint32_t)(exps[1] + 127) << 8 | (uint32_t)(exps[2] + 127) << 16 | f.n_slots << 24;
    r.order = order_bits;
    for (int k = 0; k < 4; ++k)
    {
        if (slot[k] == RT_EMPTY_REF)
        {
            r.ref[k] = RT_EMPTY_REF;
            for (int a = 0; a < 3; ++a) r.lo[a] |= 255u << (8 * k);       // lo 255 > hi 0: never hit
            continue;
        }
        const rt_bvh_node c = nodes[slot[k]];
        r.ref[k] = (c.num_primitives_axis >> 16) != 0u ? (RT_LEAF_BIT | c.offset) : s.is_root[slot[k]];
        const float cmin[3] = {c.bounds_min.x, c.bounds_min.y, c.bounds_min.z};
        const float cmax[3] = {c.bounds_max.x, c.bounds_max.y, c.bounds_max.z};
        for (int a = 0; a < 3; ++a)
        {
            const double cell = ldexp(1.0, exps[a]);
            double lo = floor(((double)cmin[a] - (double)origin[a]) / cell);
            double hi = ceil(((double)cmax[a] - (double)origin[a]) / cell);
            while ((double)origin[a] + lo * cell > (double)cmin[a]) lo -= 1.0;     // containment settled on the grid points themselves
            while ((double)origin[a] + hi * cell < (double)cmax[a]) hi += 1.0;
            if (lo < 0.0 || hi > 255.0 || lo > hi) { *s.error = FOLD_FRAME; return; }
            r.lo[a] |= (uint32_t)lo << (8 * k);
            r.hi[a] |= (uint32_t)hi << (8 * k);
        }
    }
    out[w] = r;
}

// counts[n] += rays whose slab test of binary-tree node n passes within [0, o.w] (plain binary32: a weight, not a result) -- the host's
// count_box_passes, one thread per probe ray, a 64-entry stack per thread (deeper subtrees are left out and counted in *truncated)
__global__ __launch_bounds__(64) void k_count_box_passes(const rt_bvh_node* __restrict__ nodes, uint32_t nn, const float4* __restrict__ o, const float4* __restrict__ d,
    uint32_t n_rays, uint32_t* __restrict__ counts, uint32_t* __restrict__ truncated)
{
    const uint32_t r = blockIdx.x * 64u + threadIdx.x;
    if (r >= n_rays) return;
    const float4 ro = o[r], rd = d[r];
    const float org[3] = {ro.x, ro.y, ro.z}, inv[3] = {1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z};
    const float t_max = ro.w;
    uint32_t stack[64];
    int sp = 0;
    stack[sp++] = 0;
    while (sp > 0)
    {
        const uint32_t n = stack[--sp];
        const rt_bvh_node b = nodes[n];
        const float mn[3] = {b.bounds_min.x, b.bounds_min.y, b.bounds_min.z}, mx[3] = {b.bounds_max.x, b.bounds_max.y, b.bounds_max.z};
        float t0 = 0.0f, t1 = t_max;
        for (int a = 0; a < 3; ++a)
        {
            const float ta = (mn[a] - org[a]) * inv[a], tb = (mx[a] - org[a]) * inv[a];
            t0 = __builtin_fmaxf(t0, __builtin_fminf(ta, tb));       // fmin / fmax drop a NaN (0 * inf): conservative, like the kernels
            t1 = __builtin_fminf(t1, __builtin_fmaxf(ta, tb));
        }
        if (!(t0 <= t1)) continue;
        atomicAdd(&counts[n], 1u);
        if ((b.num_primitives_axis >> 16) != 0u) continue;
        if (sp > 61) { atomicAdd(truncated, 1u); continue; }
        if (b.offset >= nn || n + 1u >= nn) continue;
        stack[sp++] = b.offset;
        stack[sp++] = n + 1u;
    }
}
