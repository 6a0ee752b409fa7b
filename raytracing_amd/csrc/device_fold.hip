// device_fold.hip -- the backend's tree work ON THE DEVICE (round 6): the collapse of a binary BVH into the 4-wide records of k_trace_w4
// and the crossing counts of a fold adaptation (kernels: fold_kernels.h), with the host side that drives them.  A translation unit -- and a device code
// object -- of its own: the hot path's kernels (rt_hip.hip) are not rebuilt, re-linked or re-hashed when anything here changes, and vice versa.
// Every function leaves the HIP error state clean and reports failure by its return value; the callers fall back to wide_bvh.cpp's build_wide_bvh,
// which is also the tests' oracle for this path (tests/test_gpu_device_fold.py: record for record).
#include <hip/hip_runtime.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include "device_fold.h"
#include "fold_kernels.h"
#include "ploc_kernels.h"
#include <hipcub/hipcub.hpp>

namespace devfold
{
struct Buffers
{
    std::vector<void*> all;
    ~Buffers() { for (void* p : all) if (p) (void)hipFree(p); }
    template <class T> bool get(T*& out, size_t count)
    {
        void* p = nullptr;
        if (hipMalloc(&p, count * sizeof(T) + 16) != hipSuccess) { (void)hipGetLastError(); out = nullptr; return false; }
        all.push_back(p);
        out = (T*)p;
        return true;
    }
    void release(void* p) { for (void*& q : all) if (q == p) q = nullptr; }     // ownership goes to the caller
};

static FoldMetric metric_of(const ownbvh::Metric* m)
{
    FoldMetric f;
    memset(&f, 0, sizeof(f));
    f.n_dirs = -1;
    if (!m) return f;
    f.iso = m->iso;
    f.n_dirs = (int)std::min<size_t>(m->dirs.size(), 8);
    for (int i = 0; i < f.n_dirs; ++i) for (int a = 0; a < 3; ++a) f.dirs[i][a] = m->dirs[(size_t)i][a];
    return f;
}

// The fold of d_nodes[nn] (device, reference layout) on `stream`.  On success: *d_records = a device array of *n_records WideNode (the caller
// frees it), entry_ref, and -- when asked for -- the records and the BVH2 node each one tests on the host.  false: the tree does not qualify
// (exactly build_wide_bvh's conditions), the metric has more than 8 directions, a device allocation failed or `cancel` was raised.
bool fold(hipStream_t stream, const rt_bvh_node* d_nodes, uint32_t nn, const rt_bvh_node& root_node, const ownbvh::Metric* metric, const double* host_weights,
    WideNode** d_records, uint32_t* n_records, uint32_t* entry_ref, std::vector<uint32_t>* roots_out, std::vector<WideNode>* records_out,
    const std::atomic<bool>* cancel, double* seconds)
{
    const auto t0 = std::chrono::steady_clock::now();
    *d_records = nullptr; *n_records = 0; *entry_ref = 0;
    if (roots_out) roots_out->clear();
    if (records_out) records_out->clear();
    if ((root_node.num_primitives_axis >> 16) != 0u) { *entry_ref = RT_LEAF_BIT | root_node.offset; return true; }     // a leaf root: no records
    if (metric && metric->dirs.size() > 8u) return false;
    if (nn < 3u) return false;
    auto cancelled = [&]() { return cancel && cancel->load(std::memory_order_relaxed); };
    Buffers buf;
    FoldState s;
    memset(&s, 0, sizeof(s));
    s.nodes = d_nodes; s.nn = nn;
    s.metric = metric_of(metric);
    double* d_weights = nullptr;
    uint32_t *frontier[2] = {nullptr, nullptr}, *d_counts = nullptr, *block_sums = nullptr, *d_roots = nullptr;
    const uint32_t scan_blocks = (nn + FOLD_SCAN_BLOCK - 1u) / FOLD_SCAN_BLOCK;
    const uint32_t max_records = nn / 2u + 1u;
    bool ok = buf.get(s.parent, nn) && buf.get(s.arrived, nn) && buf.get(s.T, nn) && buf.get(s.F, (size_t)nn * 3u) && buf.get(s.split, (size_t)nn * 3u) &&
              buf.get(s.open, nn) && buf.get(s.is_root, nn) && buf.get(s.depth, nn) && buf.get(s.error, 1) && buf.get(frontier[0], max_records) &&
              buf.get(frontier[1], max_records) && buf.get(d_counts, 4) && buf.get(block_sums, scan_blocks + 1u) && buf.get(d_roots, max_records);
    if (ok && host_weights) ok = buf.get(d_weights, nn) && hipMemcpyAsync(d_weights, host_weights, (size_t)nn * sizeof(double), hipMemcpyHostToDevice, stream) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); return false; }
    s.weights = d_weights;
    const uint32_t node_blocks = (nn + 255u) / 256u;
    ok = hipMemsetAsync(s.error, 0, sizeof(int), stream) == hipSuccess && hipMemsetAsync(s.open, 0, nn, stream) == hipSuccess &&
         hipMemsetAsync(s.split, 0, (size_t)nn * 3u, stream) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); return false; }
    hipLaunchKernelGGL(k_fold_prepare, dim3(node_blocks), dim3(256), 0, stream, s);
    hipLaunchKernelGGL(k_fold_dp, dim3(node_blocks), dim3(256), 0, stream, s);
    // top-down: the record roots, level by level (<= 33 levels: three pending slots per level must fit the kernel's stack)
    const uint32_t zero = 0;
    uint32_t n_front = 1;
    ok = hipMemcpyAsync(frontier[0], &zero, sizeof(uint32_t), hipMemcpyHostToDevice, stream) == hipSuccess;
    int err = FOLD_OK;
    for (uint32_t level = 1; ok && n_front != 0u; ++level)
    {
        if (level > 33u) { err = FOLD_TOO_DEEP; break; }
        if (cancelled()) { ok = false; break; }
        uint32_t* cur = frontier[(level - 1u) & 1u]; uint32_t* nxt = frontier[level & 1u];
        ok = hipMemsetAsync(d_counts, 0, sizeof(uint32_t), stream) == hipSuccess;
        if (!ok) break;
        hipLaunchKernelGGL(k_fold_roots, dim3((n_front + 255u) / 256u), dim3(256), 0, stream, s, (const uint32_t*)cur, n_front, nxt, d_counts, level);
        uint32_t got[2] = {0, 0};
        ok = hipMemcpyAsync(&got[0], d_counts, sizeof(uint32_t), hipMemcpyDeviceToHost, stream) == hipSuccess &&
             hipMemcpyAsync(&err, s.error, sizeof(int), hipMemcpyDeviceToHost, stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess;
        if (!ok || err != FOLD_OK) break;
        if (got[0] > max_records) { err = FOLD_NOT_A_TREE; break; }
        n_front = got[0];
    }
    if (!ok || err != FOLD_OK) { (void)hipGetLastError(); return false; }
    // record index = rank among the record roots in node order
    uint32_t total = 0;
    hipLaunchKernelGGL(k_fold_scan_sums, dim3(scan_blocks), dim3(256), 0, stream, (const uint32_t*)s.is_root, nn, block_sums);
    hipLaunchKernelGGL(k_fold_scan_blocks, dim3(1), dim3(1024), 0, stream, block_sums, scan_blocks, d_counts + 1);
    hipLaunchKernelGGL(k_fold_scan_apply, dim3(scan_blocks), dim3(256), 0, stream, s.is_root, nn, (const uint32_t*)block_sums, d_roots);
    ok = hipMemcpyAsync(&total, d_counts + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess;
    if (!ok || total == 0u || total > max_records || total >= (1u << 26)) { (void)hipGetLastError(); return false; }      // 32-bit byte offsets in the kernel
    WideNode* recs = nullptr;
    if (!buf.get(recs, total)) return false;
    hipLaunchKernelGGL(k_fold_emit, dim3((total + 63u) / 64u), dim3(64), 0, stream, s, (const uint32_t*)d_roots, total, recs);
    ok = hipMemcpyAsync(&err, s.error, sizeof(int), hipMemcpyDeviceToHost, stream) == hipSuccess;
    if (ok && roots_out) { roots_out->resize(total); ok = hipMemcpyAsync(roots_out->data(), d_roots, (size_t)total * sizeof(uint32_t), hipMemcpyDeviceToHost, stream) == hipSuccess; }
    if (ok && records_out) { records_out->resize(total); ok = hipMemcpyAsync(records_out->data(), recs, (size_t)total * sizeof(WideNode), hipMemcpyDeviceToHost, stream) == hipSuccess; }
    ok = ok && hipStreamSynchronize(stream) == hipSuccess;
    if (!ok || err != FOLD_OK || cancelled())
    {
        (void)hipGetLastError();
        if (roots_out) roots_out->clear();
        if (records_out) records_out->clear();
        return false;
    }
    buf.release(recs);
    *d_records = recs; *n_records = total; *entry_ref = 0;
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return true;
}

// ---- a binary tree over the reference's leaves, built on the device (ploc_kernels.h) ----
// exclusive prefix sum of 0 / 1 flags over n elements with fold_kernels.h's three scan kernels: flags[i] becomes its rank (RT_EMPTY_REF where the flag was 0), list[rank] = i
static bool scan_flags(hipStream_t stream, uint32_t* flags, uint32_t n, uint32_t* block_sums, uint32_t* d_total, uint32_t* list, uint32_t* total)
{
    const uint32_t blocks = (n + FOLD_SCAN_BLOCK - 1u) / FOLD_SCAN_BLOCK;
    hipLaunchKernelGGL(k_fold_scan_sums, dim3(blocks), dim3(256), 0, stream, (const uint32_t*)flags, n, block_sums);
    hipLaunchKernelGGL(k_fold_scan_blocks, dim3(1), dim3(1024), 0, stream, block_sums, blocks, d_total);
    hipLaunchKernelGGL(k_fold_scan_apply, dim3(blocks), dim3(256), 0, stream, flags, n, (const uint32_t*)block_sums, list);
    return hipMemcpyAsync(total, d_total, sizeof(uint32_t), hipMemcpyDeviceToHost, stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess;
}

// the frame of the Morton order (ploc_kernels.h): world axes, or (u, v, light) with the light's axis stretched
static PlocFrame ploc_frame(const rt_bvh_node& root, const float* light_dir, double stretch)
{
    PlocFrame f;
    memset(&f, 0, sizeof(f));
    double ax[3][3] = {{1.0, 0.0, 0.0}, {0.0, 1.0, 0.0}, {0.0, 0.0, 1.0}};
    if (light_dir)
    {
        const double d[3] = {light_dir[0], light_dir[1], light_dir[2]};
        const double len = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        if (len > 0.0 && std::isfinite(len))
        {
            const double w[3] = {d[0] / len, d[1] / len, d[2] / len};
            // u: perpendicular to w, from the world axis w is least aligned with; v = w x u
            int m = std::fabs(w[0]) <= std::fabs(w[1]) ? (std::fabs(w[0]) <= std::fabs(w[2]) ? 0 : 2) : (std::fabs(w[1]) <= std::fabs(w[2]) ? 1 : 2);
            double e[3] = {0.0, 0.0, 0.0}; e[m] = 1.0;
            const double dot = e[0] * w[0] + e[1] * w[1] + e[2] * w[2];
            double u[3] = {e[0] - dot * w[0], e[1] - dot * w[1], e[2] - dot * w[2]};
            const double ul = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
            for (double& x : u) x /= ul;
            const double v[3] = {w[1] * u[2] - w[2] * u[1], w[2] * u[0] - w[0] * u[2], w[0] * u[1] - w[1] * u[0]};
            for (int k = 0; k < 3; ++k) { ax[0][k] = u[k]; ax[1][k] = v[k]; ax[2][k] = w[k]; }
        }
    }
    const double corner[2][3] = {{root.bounds_min.x, root.bounds_min.y, root.bounds_min.z}, {root.bounds_max.x, root.bounds_max.y, root.bounds_max.z}};
    for (int a = 0; a < 3; ++a)
    {
        double lo = INFINITY, hi = -INFINITY;
        for (int c = 0; c < 8; ++c)
        {
            const double p = ax[a][0] * corner[c & 1][0] + ax[a][1] * corner[(c >> 1) & 1][1] + ax[a][2] * corner[(c >> 2) & 1][2];
            lo = std::min(lo, p); hi = std::max(hi, p);
        }
        for (int k = 0; k < 3; ++k) f.axis[a][k] = ax[a][k];
        f.lo[a] = lo; f.hi[a] = hi;
    }
    // cells of the same length along every axis (the scene's longest extent sets it), the light's axis `stretch` times longer: its interval is widened accordingly
    double longest = 0.0;
    for (int a = 0; a < 3; ++a) longest = std::max(longest, f.hi[a] - f.lo[a]);
    for (int a = 0; a < 3; ++a)
    {
        const double want = longest * (a == 2 && light_dir ? stretch : 1.0);
        f.hi[a] = f.lo[a] + (want > 0.0 ? want : 1.0);
    }
    return f;
}

bool build_tree(hipStream_t stream, const rt_bvh_node* d_ref_nodes, uint32_t nn, const rt_bvh_node& root_node, const ownbvh::Metric* metric, rt_bvh_node** d_tree, uint32_t* n_tree,
    std::vector<rt_bvh_node>* tree_out, const std::atomic<bool>* cancel, double* seconds, uint32_t* rounds_out, const float* light_dir, uint32_t radius, double stretch)
{
    if (radius == 0u) radius = PLOC_RADIUS;
    if (radius > 256u) radius = 256u;
    const auto t0 = std::chrono::steady_clock::now();
    *d_tree = nullptr; *n_tree = 0;
    if (tree_out) tree_out->clear();
    if ((root_node.num_primitives_axis >> 16) != 0u || nn < 3u) return false;                  // a leaf root: nothing to build
    if (metric && metric->dirs.size() > 8u) return false;
    auto cancelled = [&]() { return cancel && cancel->load(std::memory_order_relaxed); };
    const FoldMetric fm = metric_of(metric);
    Buffers buf;
    uint32_t *flags = nullptr, *leaf_nodes = nullptr, *block_sums = nullptr, *d_counts = nullptr, *values = nullptr, *values_sorted = nullptr;
    unsigned long long *keys = nullptr, *keys_sorted = nullptr;
    int* d_err = nullptr;
    const uint32_t scan_blocks = (nn + FOLD_SCAN_BLOCK - 1u) / FOLD_SCAN_BLOCK;
    bool ok = buf.get(flags, nn) && buf.get(leaf_nodes, nn) && buf.get(block_sums, scan_blocks + 1u) && buf.get(d_counts, 4) && buf.get(d_err, 1);
    if (!ok) return false;
    ok = hipMemsetAsync(d_err, 0, sizeof(int), stream) == hipSuccess;
    hipLaunchKernelGGL(k_ploc_flag_leaves, dim3((nn + 255u) / 256u), dim3(256), 0, stream, d_ref_nodes, nn, flags);
    uint32_t n_leaves = 0;
    ok = ok && scan_flags(stream, flags, nn, block_sums, d_counts, leaf_nodes, &n_leaves);
    if (!ok || n_leaves < 2u || 2ull * n_leaves - 1u > 0x7FFFFFFFull) { (void)hipGetLastError(); return false; }
    const uint32_t n_nodes = 2u * n_leaves - 1u;
    // Morton order of the leaves
    ok = buf.get(keys, n_leaves) && buf.get(keys_sorted, n_leaves) && buf.get(values, n_leaves) && buf.get(values_sorted, n_leaves);
    if (!ok) return false;
    const uint32_t leaf_blocks = (n_leaves + 255u) / 256u;
    hipLaunchKernelGGL(k_ploc_keys, dim3(leaf_blocks), dim3(256), 0, stream, d_ref_nodes, (const uint32_t*)leaf_nodes, n_leaves, ploc_frame(root_node, light_dir, stretch > 0.0 ? stretch : 1.0), keys, values);
    size_t temp_bytes = 0;
    void* temp = nullptr;
    if (hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, keys, keys_sorted, values, values_sorted, (int)n_leaves, 0, 63, stream) != hipSuccess) { (void)hipGetLastError(); return false; }
    { char* t = nullptr; if (!buf.get(t, temp_bytes + 256u)) return false; temp = t; }
    if (hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys, keys_sorted, values, values_sorted, (int)n_leaves, 0, 63, stream) != hipSuccess) { (void)hipGetLastError(); return false; }
    // the pool of nodes (leaves first, in Morton order) and the cluster lists
    PlocNode* pool = nullptr;
    uint32_t *parent = nullptr, *cluster[2] = {nullptr, nullptr}, *nearest = nullptr, *keep = nullptr, *merge = nullptr, *keep_list = nullptr, *merge_list = nullptr, *block_sums2 = nullptr;
    ok = buf.get(pool, n_nodes) && buf.get(parent, n_nodes) && buf.get(cluster[0], n_leaves) && buf.get(cluster[1], n_leaves) && buf.get(nearest, n_leaves) && buf.get(keep, n_leaves) &&
         buf.get(merge, n_leaves) && buf.get(keep_list, n_leaves) && buf.get(merge_list, n_leaves) && buf.get(block_sums2, (n_leaves + FOLD_SCAN_BLOCK - 1u) / FOLD_SCAN_BLOCK + 1u);
    if (!ok) return false;
    hipLaunchKernelGGL(k_ploc_init, dim3(leaf_blocks), dim3(256), 0, stream, d_ref_nodes, (const uint32_t*)leaf_nodes, (const uint32_t*)values_sorted, n_leaves, pool, parent, cluster[0]);
    uint32_t n = n_leaves, next_id = n_leaves, rounds = 0;
    int cur = 0;
    while (n > 1u)
    {
        if (++rounds > 400u || cancelled()) { (void)hipGetLastError(); return false; }       // (every round merges at least the globally cheapest pair; typical: ~30 rounds)
        const uint32_t blocks = (n + 255u) / 256u;
        hipLaunchKernelGGL(k_ploc_nearest, dim3(blocks), dim3(256), 0, stream, (const PlocNode*)pool, (const uint32_t*)cluster[cur], n, fm, radius, nearest);
        hipLaunchKernelGGL(k_ploc_decide, dim3(blocks), dim3(256), 0, stream, (const uint32_t*)nearest, n, keep, merge);
        // positions of the survivors and ids of the new nodes, both in position order (the scan turns a flag into its rank: the flags themselves are needed again, so copies are scanned)
        uint32_t n_keep = 0, n_merge = 0;
        ok = hipMemcpyAsync(keep_list, keep, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream) == hipSuccess &&
             hipMemcpyAsync(merge_list, merge, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream) == hipSuccess;
        // (scan_flags writes rank-or-EMPTY back into its flag array and the flagged indices into `list`: here the list is scratch -- nearest is still needed, so flags is reused)
        ok = ok && scan_flags(stream, keep_list, n, block_sums2, d_counts, flags, &n_keep) && scan_flags(stream, merge_list, n, block_sums2, d_counts + 1, flags, &n_merge);
        if (!ok || n_merge == 0u || n_keep + n_merge != n || next_id + n_merge > n_nodes) { (void)hipGetLastError(); return false; }
        hipLaunchKernelGGL(k_ploc_apply, dim3(blocks), dim3(256), 0, stream, pool, parent, (const uint32_t*)cluster[cur], (const uint32_t*)nearest, (const uint32_t*)keep, (const uint32_t*)merge,
            (const uint32_t*)keep_list, (const uint32_t*)merge_list, n, next_id, cluster[cur ^ 1]);
        next_id += n_merge;
        n = n_keep;
        cur ^= 1;
    }
    if (next_id != n_nodes) { (void)hipGetLastError(); return false; }
    const uint32_t root = n_nodes - 1u;                                     // the last node made
    // sizes, then the reference's linear layout
    uint32_t *arrived = nullptr, *size = nullptr, *index_of = nullptr;
    rt_bvh_node* out = nullptr;
    ok = buf.get(arrived, n_nodes) && buf.get(size, n_nodes) && buf.get(index_of, n_nodes) && buf.get(out, n_nodes);
    ok = ok && hipMemsetAsync(arrived, 0, (size_t)n_nodes * sizeof(uint32_t), stream) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); return false; }
    hipLaunchKernelGGL(k_ploc_sizes, dim3(leaf_blocks), dim3(256), 0, stream, (const PlocNode*)pool, (const uint32_t*)parent, n_leaves, n_nodes, arrived, size, d_err);
    hipLaunchKernelGGL(k_ploc_emit, dim3((n_nodes + 255u) / 256u), dim3(256), 0, stream, d_ref_nodes, (const PlocNode*)pool, (const uint32_t*)parent, (const uint32_t*)size, n_nodes, root, out, index_of, d_err);
    int err = 0;
    ok = hipMemcpyAsync(&err, d_err, sizeof(int), hipMemcpyDeviceToHost, stream) == hipSuccess;
    if (ok && tree_out) { tree_out->resize(n_nodes); ok = hipMemcpyAsync(tree_out->data(), out, (size_t)n_nodes * sizeof(rt_bvh_node), hipMemcpyDeviceToHost, stream) == hipSuccess; }
    ok = ok && hipStreamSynchronize(stream) == hipSuccess;
    if (!ok || err != FOLD_OK || cancelled()) { (void)hipGetLastError(); if (tree_out) tree_out->clear(); return false; }
    buf.release(out);
    *d_tree = out; *n_tree = n_nodes;
    if (rounds_out) *rounds_out = rounds;
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return true;
}

// count_box_passes on the device: counts[n] = probe rays whose slab test of node n passes (FoldAdapt).  The tree is uploaded here (the adaptation's
// trees live on the host: the shadow rays' one changes with every rotation).
bool count_box_passes(hipStream_t stream, const rt_bvh_node* d_nodes, uint32_t nn, const float4* o, const float4* d, size_t n_rays, std::vector<uint32_t>& counts, uint64_t* truncated)
{
    counts.assign(nn, 0u);
    if (n_rays == 0 || n_rays > 0x7FFFFFFFull) return n_rays == 0;
    Buffers buf;
    float4 *d_o = nullptr, *d_d = nullptr; uint32_t *d_counts = nullptr, *d_trunc = nullptr;
    bool ok = buf.get(d_o, n_rays) && buf.get(d_d, n_rays) && buf.get(d_counts, nn) && buf.get(d_trunc, 1);
    ok = ok && hipMemcpyAsync(d_o, o, n_rays * sizeof(float4), hipMemcpyHostToDevice, stream) == hipSuccess &&
         hipMemcpyAsync(d_d, d, n_rays * sizeof(float4), hipMemcpyHostToDevice, stream) == hipSuccess &&
         hipMemsetAsync(d_counts, 0, (size_t)nn * sizeof(uint32_t), stream) == hipSuccess && hipMemsetAsync(d_trunc, 0, sizeof(uint32_t), stream) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); return false; }
    hipLaunchKernelGGL(k_count_box_passes, dim3((uint32_t)((n_rays + 63u) / 64u)), dim3(64), 0, stream, d_nodes, nn, (const float4*)d_o, (const float4*)d_d, (uint32_t)n_rays, d_counts, d_trunc);
    uint32_t trunc = 0;
    ok = hipMemcpyAsync(counts.data(), d_counts, (size_t)nn * sizeof(uint32_t), hipMemcpyDeviceToHost, stream) == hipSuccess &&
         hipMemcpyAsync(&trunc, d_trunc, sizeof(uint32_t), hipMemcpyDeviceToHost, stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); return false; }
    if (truncated) *truncated += trunc;
    return true;
}
} // namespace devfold
