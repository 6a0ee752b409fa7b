// debug_exports_impl.h -- part of rt_hip.hip's translation unit (included inside its extern "C" block): the rt_debug_* entry points that expose the
// backend's tree work on its own (folds, own trees, tree choice, adaptation halves, pair layout, the device fold) to the tests and tools.
#pragma once

int rt_debug_wide_bvh(const rt_bvh_node* nodes, uint32_t num_nodes, int collapse, void* records, uint32_t* roots, uint32_t capacity,
    uint32_t* num_records, uint32_t* entry_ref)
{
    if (!nodes || num_nodes == 0 || !num_records || !entry_ref) return fail(nullptr, "rt_debug_wide_bvh: NULL argument");
    std::vector<WideNode> wide;
    uint32_t entry = 0;
    std::vector<uint32_t> folded;
    if (!build_wide_bvh(nodes, num_nodes, collapse == 2 ? RT_WIDE_TWO_LEVELS : RT_WIDE_SAH, wide, entry, &folded))
        return fail(nullptr, "rt_debug_wide_bvh: the tree does not qualify for the 4-wide layout (bounds not finite / not nested, or too deep)");
    *num_records = (uint32_t)wide.size();
    *entry_ref = entry;
    if (records)
    {
        if (wide.size() > capacity) return fail(nullptr, "rt_debug_wide_bvh: capacity too small");
        memcpy(records, wide.data(), wide.size() * sizeof(WideNode));
        if (roots) memcpy(roots, folded.data(), folded.size() * sizeof(uint32_t));
    }
    return RT_OK;
}

const char* rt_scene_tree_report(rt_ctx* ctx) { return ctx ? ctx->scene.tree_report.c_str() : ""; }

int rt_debug_device_fold(rt_ctx* ctx, const rt_bvh_node* nodes, uint32_t num_nodes, double iso_weight, const float* dirs, uint32_t n_dirs, const double* weights,
    void* records, uint32_t* roots, uint32_t capacity, uint32_t* num_records, uint32_t* entry_ref, double* seconds)
{
    if (!ctx || !nodes || num_nodes == 0 || !num_records || !entry_ref) return fail(ctx, "rt_debug_device_fold: NULL argument");
    (void)hipSetDevice(ctx->device);
    ownbvh::Metric m;
    const bool with_metric = iso_weight >= 0.0;
    if (with_metric)
    {
        m.iso = iso_weight;
        for (uint32_t i = 0; i < n_dirs && dirs; ++i) m.dirs.push_back({std::fabs((double)dirs[3 * i]), std::fabs((double)dirs[3 * i + 1]), std::fabs((double)dirs[3 * i + 2])});
    }
    void* d_nodes = nullptr;
    if (dev_alloc_copy(ctx, &d_nodes, nodes, (size_t)num_nodes * sizeof(rt_bvh_node)) != RT_OK) return RT_ERROR;
    WideNode* d_recs = nullptr;
    std::vector<uint32_t> folded;
    std::vector<WideNode> wide;
    const bool ok = devfold::fold(ctx->stream, (const rt_bvh_node*)d_nodes, num_nodes, nodes[0], with_metric ? &m : nullptr, weights, &d_recs, num_records, entry_ref, &folded, &wide, nullptr, seconds);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_nodes);
    if (d_recs) (void)hipFree(d_recs);
    if (!ok) return fail(ctx, "rt_debug_device_fold: the tree does not qualify for the 4-wide layout, or the device path failed");
    if (records)
    {
        if (wide.size() > capacity) return fail(ctx, "rt_debug_device_fold: capacity too small");
        memcpy(records, wide.data(), wide.size() * sizeof(WideNode));
        if (roots) memcpy(roots, folded.data(), folded.size() * sizeof(uint32_t));
    }
    return RT_OK;
}

int rt_debug_device_tree(rt_ctx* ctx, const rt_bvh_node* nodes, uint32_t num_nodes, double iso_weight, const float* dirs, uint32_t n_dirs, rt_bvh_node* out_nodes, uint32_t capacity,
    uint32_t* num_out, double* seconds, uint32_t* rounds, uint32_t radius, const float* frame_dir, double stretch)
{
    if (!ctx || !nodes || num_nodes == 0 || !num_out) return fail(ctx, "rt_debug_device_tree: NULL argument");
    (void)hipSetDevice(ctx->device);
    ownbvh::Metric m;
    m.iso = iso_weight;
    for (uint32_t i = 0; i < n_dirs && dirs; ++i) m.dirs.push_back({std::fabs((double)dirs[3 * i]), std::fabs((double)dirs[3 * i + 1]), std::fabs((double)dirs[3 * i + 2])});
    void* d_nodes = nullptr;
    if (dev_alloc_copy(ctx, &d_nodes, nodes, (size_t)num_nodes * sizeof(rt_bvh_node)) != RT_OK) return RT_ERROR;
    rt_bvh_node* d_tree = nullptr;
    uint32_t n = 0;
    std::vector<rt_bvh_node> tree;
    const bool ok = devfold::build_tree(ctx->stream, (const rt_bvh_node*)d_nodes, num_nodes, nodes[0], &m, &d_tree, &n, &tree, nullptr, seconds, rounds, frame_dir, radius, stretch);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_nodes);
    if (d_tree) (void)hipFree(d_tree);
    if (!ok) return fail(ctx, "rt_debug_device_tree: nothing to build (leaf root), or the device path failed");
    *num_out = n;
    if (out_nodes)
    {
        if (n > capacity) return fail(ctx, "rt_debug_device_tree: capacity too small");
        memcpy(out_nodes, tree.data(), tree.size() * sizeof(rt_bvh_node));
    }
    return RT_OK;
}

int rt_debug_pair_layout(const rt_bvh_node* nodes, uint32_t num_nodes, void* records, uint32_t* roots, uint32_t num_records)
{
    if (!nodes || !records || !roots || num_nodes == 0) return fail(nullptr, "rt_debug_pair_layout: NULL argument");
    std::vector<WideNode> wide((const WideNode*)records, (const WideNode*)records + num_records);
    std::vector<uint32_t> r(roots, roots + num_records);
    pair_layout_by_area(wide, r, nodes, num_nodes, (const ownbvh::Metric*)nullptr);
    memcpy(records, wide.data(), wide.size() * sizeof(WideNode));
    memcpy(roots, r.data(), r.size() * sizeof(uint32_t));
    return RT_OK;
}

int rt_debug_wide_bvh_weights(const rt_bvh_node* nodes, uint32_t num_nodes, const double* weights, void* records, uint32_t* roots, uint32_t capacity,
    uint32_t* num_records, uint32_t* entry_ref)
{
    if (!nodes || num_nodes == 0 || !weights || !num_records || !entry_ref) return fail(nullptr, "rt_debug_wide_bvh_weights: NULL argument");
    std::vector<WideNode> wide;
    std::vector<uint32_t> folded;
    uint32_t entry = 0;
    if (!build_wide_bvh(nodes, num_nodes, RT_WIDE_SAH, wide, entry, &folded, nullptr, weights)) return fail(nullptr, "rt_debug_wide_bvh_weights: the tree does not qualify for the 4-wide layout");
    *num_records = (uint32_t)wide.size();
    *entry_ref = entry;
    if (records)
    {
        if (wide.size() > capacity) return fail(nullptr, "rt_debug_wide_bvh_weights: capacity too small");
        memcpy(records, wide.data(), wide.size() * sizeof(WideNode));
        if (roots) memcpy(roots, folded.data(), folded.size() * sizeof(uint32_t));
    }
    return RT_OK;
}

int rt_debug_choose_tree(const rt_scene_desc* sd, int shadow, uint32_t mode, void* records, uint32_t capacity, uint32_t* num_records,
    uint32_t* entry_ref, char* report, size_t report_len)
{
    if (!sd || !sd->nodes || !sd->triangles || !num_records || !entry_ref) return fail(nullptr, "rt_debug_choose_tree: NULL argument");
    std::vector<WideNode> ref_wide;
    uint32_t ref_entry = 0;
    if (!build_wide_bvh(sd->nodes, sd->num_nodes, RT_WIDE_SAH, ref_wide, ref_entry) || ref_wide.empty())
        return fail(nullptr, "rt_debug_choose_tree: the tree does not qualify for the 4-wide layout");
    std::string rep;
    OwnTree own;
    own.start(sd, shadow != 0, mode);
    const bool picked = choose_tree(sd, ref_wide, ref_entry, shadow != 0, mode, own, rep);
    const std::vector<WideNode>& w = picked ? own.wide : ref_wide;
    *num_records = (uint32_t)w.size();
    *entry_ref = picked ? own.entry : ref_entry;
    if (report && report_len) snprintf(report, report_len, "%s", rep.c_str());
    if (records)
    {
        if (w.size() > capacity) return fail(nullptr, "rt_debug_choose_tree: capacity too small");
        memcpy(records, w.data(), w.size() * sizeof(WideNode));
    }
    return RT_OK;
}

int rt_debug_own_bvh(const rt_bvh_node* nodes, uint32_t num_nodes, double iso_weight, const float* dirs, uint32_t n_dirs,
    rt_bvh_node* out_nodes, uint32_t capacity, uint32_t* num_out)
{
    if (!nodes || num_nodes == 0 || !num_out) return fail(nullptr, "rt_debug_own_bvh: NULL argument");
    ownbvh::Metric m;
    m.iso = iso_weight;
    for (uint32_t i = 0; i < n_dirs && dirs; ++i) m.dirs.push_back({std::fabs((double)dirs[3 * i]), std::fabs((double)dirs[3 * i + 1]), std::fabs((double)dirs[3 * i + 2])});
    std::vector<rt_bvh_node> own;
    if (!ownbvh::build(nodes, num_nodes, m, own)) return fail(nullptr, "rt_debug_own_bvh: nothing to build (leaf root) or the node array is not a tree");
    *num_out = (uint32_t)own.size();
    if (out_nodes)
    {
        if (own.size() > capacity) return fail(nullptr, "rt_debug_own_bvh: capacity too small");
        memcpy(out_nodes, own.data(), own.size() * sizeof(rt_bvh_node));
    }
    return RT_OK;
}

int rt_debug_wide_bvh_metric(const rt_bvh_node* nodes, uint32_t num_nodes, double iso_weight, const float* dirs, uint32_t n_dirs,
    void* records, uint32_t capacity, uint32_t* num_records, uint32_t* entry_ref)
{
    if (!nodes || num_nodes == 0 || !num_records || !entry_ref) return fail(nullptr, "rt_debug_wide_bvh_metric: NULL argument");
    ownbvh::Metric m;
    m.iso = iso_weight;
    for (uint32_t i = 0; i < n_dirs && dirs; ++i) m.dirs.push_back({std::fabs((double)dirs[3 * i]), std::fabs((double)dirs[3 * i + 1]), std::fabs((double)dirs[3 * i + 2])});
    std::vector<WideNode> wide;
    uint32_t entry = 0;
    if (!build_wide_bvh(nodes, num_nodes, RT_WIDE_SAH, wide, entry, nullptr, &m))
        return fail(nullptr, "rt_debug_wide_bvh_metric: the tree does not qualify for the 4-wide layout");
    *num_records = (uint32_t)wide.size();
    *entry_ref = entry;
    if (records)
    {
        if (wide.size() > capacity) return fail(nullptr, "rt_debug_wide_bvh_metric: capacity too small");
        memcpy(records, wide.data(), wide.size() * sizeof(WideNode));
    }
    return RT_OK;
}

// RT_CTX_OPT_ADAPTIVE_FOLD's host half on its own (no device): the surface-area fold of `nodes`, then the fold adapted to `n_rays` rays
// (origin.xyz + t_max in .w, direction.xyz) -- the records of the latter, and what both cost those rays (box passes at record roots).
int rt_debug_adapt_fold(const rt_bvh_node* nodes, uint32_t num_nodes, const float* origins_tmax, const float* directions, uint32_t n_rays,
    void* records, uint32_t* roots, uint32_t capacity, uint32_t* num_records, uint32_t* entry_ref, double* cost2, int* cheaper)
{
    if (!nodes || num_nodes == 0 || !origins_tmax || !directions || !num_records || !entry_ref) return fail(nullptr, "rt_debug_adapt_fold: NULL argument");
    std::vector<rt_bvh_node> tree(nodes, nodes + num_nodes);
    std::vector<WideNode> wide, adapted;
    std::vector<uint32_t> wide_roots;
    uint32_t entry = 0;
    if (!build_wide_bvh(nodes, num_nodes, RT_WIDE_SAH, wide, entry, &wide_roots) || wide.empty())
        return fail(nullptr, "rt_debug_adapt_fold: the tree does not qualify for the 4-wide layout");
    std::vector<float4> o(n_rays), d(n_rays);
    for (uint32_t i = 0; i < n_rays; ++i)
    {
        o[i] = make_float4(origins_tmax[4 * i], origins_tmax[4 * i + 1], origins_tmax[4 * i + 2], origins_tmax[4 * i + 3]);
        d[i] = make_float4(directions[4 * i], directions[4 * i + 1], directions[4 * i + 2], 0.0f);
    }
    double cost[2] = {0.0, 0.0};
    std::atomic<bool> cancel{false};
    std::vector<uint32_t> adapted_roots;
    const bool better = refold_for_rays(tree, o, d, wide_roots, adapted, entry, cost, cancel, &adapted_roots);
    if (adapted.empty()) return fail(nullptr, "rt_debug_adapt_fold: no adapted fold (no ray passed the root box, or the weighted fold is too deep)");
    if (cost2) { cost2[0] = cost[0]; cost2[1] = cost[1]; }
    if (cheaper) *cheaper = better ? 1 : 0;
    *num_records = (uint32_t)adapted.size();
    *entry_ref = entry;
    if (records)
    {
        if (adapted.size() > capacity) return fail(nullptr, "rt_debug_adapt_fold: capacity too small");
        memcpy(records, adapted.data(), adapted.size() * sizeof(WideNode));
        if (roots) memcpy(roots, adapted_roots.data(), adapted_roots.size() * sizeof(uint32_t));
    }
    return RT_OK;
}

// RT_CTX_OPT_ADAPTIVE_FOLD's trigger on its own: has camera `now` left the view the folds were adapted to (`adapted`), in a scene of this diagonal?
int rt_debug_fold_view_left(const rt_camera* adapted, const rt_camera* now, double scene_diagonal)
{
    if (!adapted || !now) return -1;
    FoldAdapt a;
    a.camera = *adapted;
    a.scene_diagonal = scene_diagonal;
    return fold_view_left(a, *now) ? 1 : 0;
}

// FoldAdapt's shadow side exactly as the worker runs it (adapt_shadow_side; host only): `nodes` is the shadow rays' current binary tree under its surface-area
// fold, `mode` RT_CTX_OPT_ADAPTIVE_FOLD's value (bit 3 = rotate first).  Out: the candidate's records, the tree they fold (out_tree[num_nodes]; `nodes`
// again when nothing was rotated), cost2 = {current, candidate}, *rotations, return value 1 = would be adopted, 0 = kept, < 0 = error.
int rt_debug_adapt_shadow_side(const rt_bvh_node* nodes, uint32_t num_nodes, const float* origins_tmax, const float* directions, uint32_t n_rays, uint32_t mode,
    void* records, uint32_t* roots, uint32_t capacity, uint32_t* num_records, uint32_t* entry_ref, rt_bvh_node* out_tree, double* cost2, uint32_t* rotations,
    const rt_triangle* triangles, uint32_t num_triangles, uint32_t* reordered)
{
    if (!nodes || num_nodes == 0 || !origins_tmax || !directions || !num_records || !entry_ref) { fail(nullptr, "rt_debug_adapt_shadow_side: NULL argument"); return -1; }
    FoldAdapt a;
    a.mode = mode;
    if (triangles && (mode & 16u))
    {
        a.tri9.resize((size_t)num_triangles * 9);
        for (uint32_t i = 0; i < num_triangles; ++i)
        {
            const rt_float3 v[3] = {triangles[i].v1.position, triangles[i].v2.position, triangles[i].v3.position};
            for (int k = 0; k < 3; ++k) { a.tri9[(size_t)i * 9 + 3 * k] = v[k].x; a.tri9[(size_t)i * 9 + 3 * k + 1] = v[k].y; a.tri9[(size_t)i * 9 + 3 * k + 2] = v[k].z; }
        }
    }
    a.bvh2.assign(nodes, nodes + num_nodes);
    std::vector<WideNode> wide;
    uint32_t entry = 0;
    if (!build_wide_bvh(nodes, num_nodes, RT_WIDE_SAH, wide, entry, &a.roots) || wide.empty()) { fail(nullptr, "rt_debug_adapt_shadow_side: the tree does not qualify for the 4-wide layout"); return -1; }
    a.sh_o.resize(n_rays); a.sh_d.resize(n_rays);
    for (uint32_t i = 0; i < n_rays; ++i)
    {
        a.sh_o[i] = make_float4(origins_tmax[4 * i], origins_tmax[4 * i + 1], origins_tmax[4 * i + 2], origins_tmax[4 * i + 3]);
        a.sh_d[i] = make_float4(directions[4 * i], directions[4 * i + 1], directions[4 * i + 2], 0.0f);
    }
    const bool adopted = adapt_shadow_side(&a);
    if (a.wide_sh.empty()) { fail(nullptr, "rt_debug_adapt_shadow_side: no candidate (no ray passed the root box, or the folds are too deep)"); return -1; }
    *num_records = (uint32_t)a.wide_sh.size();
    *entry_ref = a.entry_sh;
    if (cost2) { cost2[0] = a.cost[1][0]; cost2[1] = a.cost[1][1]; }
    if (rotations) *rotations = a.rotations;
    if (reordered) *reordered = a.reordered;
    if (records)
    {
        if (a.wide_sh.size() > capacity) { fail(nullptr, "rt_debug_adapt_shadow_side: capacity too small"); return -1; }
        memcpy(records, a.wide_sh.data(), a.wide_sh.size() * sizeof(WideNode));
        if (roots) memcpy(roots, a.roots_sh_new.data(), a.roots_sh_new.size() * sizeof(uint32_t));
    }
    if (out_tree) memcpy(out_tree, a.rotations != 0 ? a.bvh2_sh_new.data() : nodes, (size_t)num_nodes * sizeof(rt_bvh_node));
    return adopted ? 1 : 0;
}

// What rt_scene_upload / rt_ctx_destroy do to an adaptation in flight (host only): a FoldAdapt whose worker has just started on `nodes` and the rays
// given is dropped after delay_ms; returns the milliseconds the drop took (the worker gives up at its next check), -1 on an argument error.
double rt_debug_fold_abandon(const rt_bvh_node* nodes, uint32_t num_nodes, const float* origins_tmax, const float* directions, uint32_t n_rays, uint32_t mode,
    uint32_t delay_ms, int* had_finished)
{
    if (!nodes || num_nodes == 0 || !origins_tmax || !directions) { fail(nullptr, "rt_debug_fold_abandon: NULL argument"); return -1.0; }
    FoldAdapt* a = new FoldAdapt();
    a->mode = mode;
    a->bvh2.assign(nodes, nodes + num_nodes);
    std::vector<WideNode> wide;
    uint32_t entry = 0;
    if (!build_wide_bvh(nodes, num_nodes, RT_WIDE_SAH, wide, entry, &a->roots) || wide.empty()) { delete a; fail(nullptr, "rt_debug_fold_abandon: the tree does not qualify"); return -1.0; }
    a->o.resize(n_rays); a->d.resize(n_rays);
    for (uint32_t i = 0; i < n_rays; ++i)
    {
        a->o[i] = make_float4(origins_tmax[4 * i], origins_tmax[4 * i + 1], origins_tmax[4 * i + 2], origins_tmax[4 * i + 3]);
        a->d[i] = make_float4(directions[4 * i], directions[4 * i + 1], directions[4 * i + 2], 0.0f);
    }
    a->sh_o = a->o; a->sh_d = a->d;
    a->state = FoldAdapt::COMPUTING;
    a->worker = std::thread(fold_adapt_worker, a);
    std::this_thread::sleep_for(std::chrono::milliseconds(delay_ms));
    if (had_finished) *had_finished = a->finished.load() ? 1 : 0;
    const auto t0 = std::chrono::steady_clock::now();
    drop_fold_adapt(a);
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// tree_rotate.h on its own (host only): the binary tree `nodes` rotated for the rays given (as rt_debug_adapt_fold takes them); out_nodes[num_nodes]
int rt_debug_rotate_tree(const rt_bvh_node* nodes, uint32_t num_nodes, const float* origins_tmax, const float* directions, uint32_t n_rays, int max_passes,
    rt_bvh_node* out_nodes, double* cost2, uint32_t* rotations, int moves, double min_gain)
{
    if (!nodes || num_nodes == 0 || !origins_tmax || !directions || !out_nodes) return fail(nullptr, "rt_debug_rotate_tree: NULL argument");
    std::vector<rt_bvh_node> out;
    double cost[2] = {0.0, 0.0};
    const uint32_t made = treerot::rotate(nodes, num_nodes, origins_tmax, directions, n_rays, max_passes, out, cost, nullptr, moves, min_gain);
    if (out.size() != num_nodes) return fail(nullptr, "rt_debug_rotate_tree: the node array is not a tree");
    memcpy(out_nodes, out.data(), out.size() * sizeof(rt_bvh_node));
    if (cost2) { cost2[0] = cost[0]; cost2[1] = cost[1]; }
    if (rotations) *rotations = made;
    return RT_OK;
}

