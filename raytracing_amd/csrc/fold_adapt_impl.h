// fold_adapt_impl.h -- part of rt_hip.hip's translation unit (included inside its anonymous namespace): which tree a ray population walks (OwnTree,
// choose_tree) and the state + worker of RT_CTX_OPT_ADAPTIVE_FOLD (FoldAdapt: crossing counts, re-fold, tree rotations, occluder-first slots, the upload
// of the adapted records).  The render-thread side -- probe, hand-over, adoption -- is fold_hooks_impl.h.  Split out of rt_hip.hip in round 6.
#pragma once

// (the host-side fold, the pair layout and the adaptation's host walks: wide_bvh.cpp; the fold on the device: device_fold.hip)
// Which tree a ray population walks: the candidate of own_bvh.h against the reference's own topology (`ref_wide`), both
// walked by proxy rays of that population (tree_select.h).  mode 1: own only if it saves more than 10 % of the steps (the proxy rays are not the
// camera's: a tree that promised 6 % fewer steps on the ShaderBalls-class scene made its shadow trace 10 % slower, profiles/r04_call01_*);
// mode 2: own whatever it costs (A/B runs); mode 3 (shadow): own with the plain surface-area metric, unconditionally (A/B).
struct OwnTree
{
    std::vector<WideNode> wide; uint32_t entry = 0; bool ok = false; const char* name = ""; std::thread worker;
    std::vector<rt_bvh_node> bvh2; std::vector<uint32_t> roots;    // the binary tree the records fold, and the node each record tests (FoldAdapt)
    WideNode* d_wide = nullptr;                                    // the records on the device already (RT_CTX_OPT_DEVICE_FOLD); whoever adopts them owns them
    int device = -1;                                               // >= 0: fold on that device (a stream of the worker's own)
    bool pairs = false;                                            // RT_CTX_OPT_WIDE_LAYOUT
    bool device_builder = false;                                   // RT_CTX_OPT_TREE_BUILDER: the binary tree itself is built on the device (PLOC, ploc_kernels.h) ...
    bool device_only = false;                                      // ... and if that fails this candidate is simply not there (another OwnTree is the host-built candidate)
    std::atomic<bool> cancel_build{false};                         // the host build gives up: another candidate has won
    bool built_on_device = false; uint32_t ploc_rounds = 0;
    // RT_CTX_OPT_TREE_BUILDER = 1: tree AND fold on the device -- the reference's nodes go up once, the tree is clustered there (devfold::build_tree), folded there
    // (devfold::fold takes the device array as it is) and comes back once for the adaptation's host side; false = something failed: the host path takes over
    bool build_and_fold_on_device(const rt_scene_desc* sd, const ownbvh::Metric& m)
    {
        if (device < 0 || hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return false; }
        hipStream_t st = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return false; }
        void* d_ref = nullptr;
        rt_bvh_node* d_tree = nullptr;
        uint32_t n_tree = 0, n = 0;
        bool done = hipMalloc(&d_ref, (size_t)sd->num_nodes * sizeof(rt_bvh_node)) == hipSuccess &&
                    hipMemcpyAsync(d_ref, sd->nodes, (size_t)sd->num_nodes * sizeof(rt_bvh_node), hipMemcpyHostToDevice, st) == hipSuccess;
        const auto t0 = std::chrono::steady_clock::now();
        done = done && devfold::build_tree(st, (const rt_bvh_node*)d_ref, sd->num_nodes, sd->nodes[0], &m, &d_tree, &n_tree, &bvh2, nullptr, nullptr, &ploc_rounds);
        build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const auto t1 = std::chrono::steady_clock::now();
        done = done && devfold::fold(st, d_tree, n_tree, bvh2[0], &m, nullptr, &d_wide, &n, &entry, &roots, &wide) && n != 0u;
        (void)hipStreamSynchronize(st);
        if (d_ref) (void)hipFree(d_ref);
        if (d_tree) (void)hipFree(d_tree);
        (void)hipStreamDestroy(st);
        (void)hipGetLastError();
        if (!done) { if (d_wide) { (void)hipFree(d_wide); d_wide = nullptr; } bvh2.clear(); wide.clear(); roots.clear(); return false; }
        if (pairs)
        {
            pair_layout_by_area(wide, roots, bvh2.data(), (uint32_t)bvh2.size(), &m);
            if (hipMemcpy(d_wide, wide.data(), wide.size() * sizeof(WideNode), hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(d_wide); d_wide = nullptr; return false; }
        }
        fold_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
        built_on_device = true;
        name = name_device.c_str();
        return true;
    }
    std::string name_device;
    double fold_seconds = 0.0, build_seconds = 0.0;
    // the collapse of the finished binary tree: on the device (the tree goes up, the records stay there and come back for the choice by proxy rays), or by build_wide_bvh
    bool fold_it(const ownbvh::Metric& m)
    {
        const auto t0 = std::chrono::steady_clock::now();
        bool done = false;
        if (device >= 0 && hipSetDevice(device) == hipSuccess)
        {
            hipStream_t st = nullptr;
            void* d_nodes = nullptr;
            if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess)
            {
                if (hipMalloc(&d_nodes, bvh2.size() * sizeof(rt_bvh_node)) == hipSuccess &&
                    hipMemcpyAsync(d_nodes, bvh2.data(), bvh2.size() * sizeof(rt_bvh_node), hipMemcpyHostToDevice, st) == hipSuccess)
                {
                    uint32_t n = 0;
                    done = devfold::fold(st, (const rt_bvh_node*)d_nodes, (uint32_t)bvh2.size(), bvh2[0], &m, nullptr, &d_wide, &n, &entry, &roots, &wide) && n != 0u;
                }
                (void)hipStreamSynchronize(st);
                if (d_nodes) (void)hipFree(d_nodes);
                (void)hipStreamDestroy(st);
            }
            (void)hipGetLastError();
            if (!done && d_wide) { (void)hipFree(d_wide); d_wide = nullptr; }
        }
        if (!done) done = build_wide_bvh(bvh2.data(), (uint32_t)bvh2.size(), RT_WIDE_SAH, wide, entry, &roots, &m) && !wide.empty();
        if (done && pairs)
        {
            pair_layout_by_area(wide, roots, bvh2.data(), (uint32_t)bvh2.size(), &m);
            if (d_wide && hipMemcpy(d_wide, wide.data(), wide.size() * sizeof(WideNode), hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(d_wide); d_wide = nullptr; }
        }
        fold_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return done;
    }
    void start(const rt_scene_desc* sd, bool shadow, uint32_t mode)
    {
        ownbvh::Metric m;
        name = "own: surface area";
        if (shadow && mode != 3u)
        {
            ownbvh::Metric d = shadow_metric(sd->lights, sd->num_lights, 0.5);
            if (!d.dirs.empty()) { m = d; name = "own: projected area along the directional lights + 50 % isotropic"; }
        }
        name_device = std::string(name) + ", built on the device (PLOC)";
        worker = std::thread([this, sd, m]()
        {
            if (device_builder && build_and_fold_on_device(sd, m)) { ok = true; return; }
            if (device_only) { ok = false; return; }
            const auto t0 = std::chrono::steady_clock::now();
            ok = ownbvh::build(sd->nodes, sd->num_nodes, m, bvh2, &cancel_build);
            build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            ok = ok && fold_it(m);
        });
    }
    void join() { if (worker.joinable()) worker.join(); }
    ~OwnTree() { join(); if (d_wide) (void)hipFree(d_wide); }
};

// What every candidate of one ray population is measured against and with: the proxy rays (an area-weighted pass over all triangles), the leaf of every first
// triangle, the reference fold's cost -- made by the first choose_tree call and kept for the next (the device-built candidate, then the host-built one).
struct ChoiceInputs
{
    bool valid = false;
    std::vector<uint32_t> leaf_of_first;
    std::vector<treesel::ProxyRay> rays;
    double c_ref = 0.0;
};

// true: walk the own tree; false: keep the reference topology
bool choose_tree(const rt_scene_desc* sd, const std::vector<WideNode>& ref_wide, uint32_t ref_entry, bool shadow, uint32_t mode,
    OwnTree& own, std::string& report, ChoiceInputs* kept = nullptr)
{
    own.join();
    char line[320];
    if (!own.ok) { report += shadow ? "shadow tree: the own tree does not qualify -> reference topology\n" : "closest-hit tree: the own tree does not qualify -> reference topology\n"; return false; }
    if (mode >= 2u)
    {
        snprintf(line, sizeof(line), "%s tree: %s, forced (not measured)\n", shadow ? "shadow" : "closest-hit", own.name);
        report += line;
        return true;
    }
    const uint32_t nt = sd->num_triangles, nn = sd->num_nodes;
    ChoiceInputs mine;
    ChoiceInputs& in = kept ? *kept : mine;
    if (!in.valid)
    {
        in.leaf_of_first.assign(nt, 0u);
        for (uint32_t i = 0; i < nn; ++i)
            if ((sd->nodes[i].num_primitives_axis >> 16) != 0 && sd->nodes[i].offset < nt) in.leaf_of_first[sd->nodes[i].offset] = i;
        in.rays = treesel::proxy_rays(sd->triangles, nt, sd->lights, sd->num_lights, 8192u, shadow);
        if (!in.rays.empty())
            in.c_ref = treesel::walk_cost((const treesel::Record*)ref_wide.data(), (uint32_t)ref_wide.size(), ref_entry, sd->nodes, in.leaf_of_first.data(), sd->triangles, in.rays, shadow);
        in.valid = true;
    }
    const std::vector<uint32_t>& leaf_of_first = in.leaf_of_first;
    const std::vector<treesel::ProxyRay>& rays = in.rays;
    if (rays.empty()) { report += shadow ? "shadow tree: no lights, nothing to measure -> reference topology\n" : "closest-hit tree: nothing to measure -> reference topology\n"; return false; }
    const double c_ref = in.c_ref;
    const double c_own = treesel::walk_cost((const treesel::Record*)own.wide.data(), (uint32_t)own.wide.size(), own.entry, sd->nodes, leaf_of_first.data(), sd->triangles, rays, shadow);
    const bool pick = c_own < 0.90 * c_ref;
    snprintf(line, sizeof(line), "%s tree: reference topology %.2f steps per proxy ray, %s %.2f -> %s\n", shadow ? "shadow" : "closest-hit", c_ref, own.name, c_own,
        pick ? "own" : "reference topology");
    report += line;
    return pick;
}

// ---- Fold adaptation (RT_CTX_OPT_ADAPTIVE_FOLD) ---------------------------------------------------------------------------------
// build_wide_bvh's dynamic programme is optimal for whatever visit probability it is given, and the surface area is only the
// probability of a ray population nobody traces: uniformly distributed lines.  The rays of a frame are not that (they start at
// the camera or on surfaces and stop at the first hit), and what they do can be measured: the first rt_integrate of a scene
// traces a small probe frame, the host counts how often its rays pass each box of the binary tree (closest-hit rays clipped at
// their hit), and the trees are folded again for those frequencies -- tools/fold_weight_study.py: - 8 % closest-hit and - 10 %
// shadow record visits on the benchmark scene, out of sample, and 14 000 probe rays are as good as 220 000.
// Exact by construction: every fold of the same binary tree tests the same leaves in the same order (build_wide_bvh).
struct FoldAdapt
{
    enum { ARMED = 1, COMPUTING = 2, IDLE = 3, OFF = 4, PROBING = 5 };   // IDLE: adapted to `camera`; a frame whose camera has moved away arms it again;
                                                                         // PROBING: the probe frame's launches and copies are on the stream
    int state = ARMED;
    std::atomic<uint32_t> mode{1};                     // ctx->adaptive_fold at upload (atomic: RT_CTX_OPT_ADAPT_WAIT changes bit 1 on the render thread while the worker reads bits 3 / 4)
    uint32_t adaptations = 0;                          // folds adopted so far
    rt_camera camera;                                  // the probe's camera
    double scene_diagonal = 0.0;
    std::vector<rt_bvh_node> bvh2, bvh2_sh;            // the reference's tree; the shadow rays' own binary tree (empty: they walk the reference's)
    std::vector<uint32_t> roots, roots_sh;             // the binary-tree node each record of the CURRENT folds tests
    std::vector<uint32_t> roots_new, roots_sh_new;     // ... of the adapted folds
    std::vector<float> tri9;                           // mode bit 4: the triangles' corner positions (9 floats each), for the host's occluder search
    uint32_t reordered = 0;                            // ... shadow records whose slots changed places (0: placed as build_wide_bvh places them)
    std::vector<rt_bvh_node> bvh2_sh_new;              // mode bit 3: the shadow rays' binary tree after tree_rotate.h's rotations (when that is what was folded)
    uint32_t rotations = 0;                            // ... how many (0: the fold is of the tree as it was)
    std::vector<float4> o, d, sh_o, sh_d;              // the probe's rays (o.w = t_max: the hit distance where there was one)
    std::vector<WideNode> wide, wide_sh;               // the adapted folds
    uint32_t entry = 0, entry_sh = 0;
    bool ok = false, ok_sh = false;                    // ... exist and are cheaper for the probe rays
    double cost[2][2] = {{0.0, 0.0}, {0.0, 0.0}};      // [closest, shadow][current, adapted]: record visits per probe ray (an upper bound: box passes)
    double seconds = 0.0;
    double rotate_s[4] = {0, 0, 0, 0};                     // ... and tree_rotate.h's phases
    double stage_s[7] = {0, 0, 0, 0, 0, 0, 0};             // the worker's stages: unpacking the probe, closest-hit re-fold, shadow: plain re-fold, rotations, rotated re-fold, occluder order; upload
    std::atomic<bool> finished{false}, cancel{false};
    std::thread worker;
    // The probe (round 5: nothing on the render thread waits for it): a frame of its own, kept for the scene's life; its queues come back through
    // pinned memory with asynchronous copies behind each stage, `probe_done` marks the last one; the worker unpacks them (probe_unpack).
    rt_frame* probe = nullptr;
    uint32_t probe_paths = 0, probe_samples = 0, probe_bounces = 0;      // capacity of a queue, samples traced, bounces + 1
    char* staging = nullptr; size_t staging_bytes = 0;                   // pinned; layout: probe_block / probe_counters below
    hipEvent_t probe_done = nullptr;
    // The device side of an adoption is the WORKER's too: it uploads the adapted records on a stream of its own, and frees the ones an
    // earlier adoption replaced after a device synchronisation of ITS thread (every launch that could still read them was enqueued before
    // that adoption).  The render thread only exchanges pointers: no hipDeviceSynchronize, no hipMalloc / hipFree between two frames.
    int device = -1;                                                     // -1: host only (rt_debug_fold_abandon)
    bool device_fold = false;                                            // RT_CTX_OPT_DEVICE_FOLD: crossing counts and re-folds on `device` ...
    // ... when the device has nothing better to do: with bit 1 (rt_integrate waits for the adaptation) it is idle and the folds on it shorten the wait
    // (1.33 -> 1.09 s to the adapted fold on the headline scene, 8.5 -> 7.1 s on the 10 M-triangle one: profiles/r06/call03.log); asynchronously -- the library's
    // default -- the frames keep it busy and the host's threads are what is free: the same folds on the device took an orbiting camera's frames from
    // + 2 .. 4 % to + 23 % (per_frame.moving_camera, profiles/r06/call04.log), so there the worker folds on host threads as it did before.
    // Measured again in round 6's call 24, with the worker's other stages short: the folds on the device beside an orbiting camera's frames take 0.36 - 0.60 s instead of the
    // host's 0.22 - 0.25 (their kernels queue behind the frames') and the frames 3.6 - 3.8 ms instead of 2.68 (+ 40 %): the policy stands.
    int worker_fold_device() const { return device_fold && (mode.load() & 2u) ? device : -1; }
    bool pairs = false;                                                  // RT_CTX_OPT_WIDE_LAYOUT
    void *new_cl = nullptr, *new_sh = nullptr;
    bool upload_failed = false;
    std::vector<void*> retired;
    std::chrono::steady_clock::time_point last_armed{};                  // re-arming is rate-limited (RT_CTX_OPT_ADAPT_MIN_INTERVAL_MS)
    std::atomic<uint32_t> min_interval_ms{500};
    size_t probe_block(uint32_t sample, uint32_t bounce, uint32_t which /* 0 o, 1 d, 2 hits, 3 shadow o, 4 shadow d */) const
    {
        return ((((size_t)sample * probe_bounces + bounce) * 5u + which) * probe_paths) * sizeof(float4);
    }
    size_t probe_counters(uint32_t sample) const { return (size_t)probe_samples * probe_bounces * 5u * probe_paths * sizeof(float4) + (size_t)sample * sizeof(DCounters); }
    ~FoldAdapt();
};
void drop_fold_adapt(FoldAdapt* a) { delete a; }
void fold_adapt_set_interval(FoldAdapt* a, uint32_t ms) { a->min_interval_ms = ms; }
void fold_adapt_set_wait(FoldAdapt* a, bool wait) { if (wait) a->mode.fetch_or(2u); else a->mode.fetch_and(~2u); }

// One tree folded again for the rays that were counted on it.  cost[] = what the current and the new fold cost those rays.
// fold_device >= 0 (RT_CTX_OPT_DEVICE_FOLD): the crossing counts and the collapse run on that device, on a stream of the calling (worker) thread's own -- the tree
// goes up once per call (the shadow rays' tree changes with every rotation), the records come back for the host's bookkeeping; anything that fails there
// is done here on host threads instead.
bool refold_for_rays(const std::vector<rt_bvh_node>& tree, const std::vector<float4>& o, const std::vector<float4>& d, const std::vector<uint32_t>& roots_now,
    std::vector<WideNode>& out, uint32_t& entry, double (&cost)[2], const std::atomic<bool>& cancel, std::vector<uint32_t>* roots_out = nullptr, int fold_device = -1,
    bool pairs = false /* RT_CTX_OPT_WIDE_LAYOUT: the new records in (parent, likeliest child) pairs, by the measured weights */)
{
    if (tree.empty() || o.empty() || o.size() != d.size() || roots_now.empty()) return false;
    const uint32_t nn = (uint32_t)tree.size();
    std::vector<uint32_t> counts;
    struct DeviceTree
    {
        hipStream_t st = nullptr; void* nodes = nullptr;
        ~DeviceTree() { if (st) (void)hipStreamSynchronize(st); if (nodes) (void)hipFree(nodes); if (st) (void)hipStreamDestroy(st); (void)hipGetLastError(); }
    } dev;
    bool on_device = false;
    if (fold_device >= 0 && hipSetDevice(fold_device) == hipSuccess && hipStreamCreateWithFlags(&dev.st, hipStreamNonBlocking) == hipSuccess)
    {
        on_device = hipMalloc(&dev.nodes, (size_t)nn * sizeof(rt_bvh_node)) == hipSuccess &&
                    hipMemcpyAsync(dev.nodes, tree.data(), (size_t)nn * sizeof(rt_bvh_node), hipMemcpyHostToDevice, dev.st) == hipSuccess;
        if (on_device)
        {
            uint64_t truncated = 0;
            on_device = devfold::count_box_passes(dev.st, (const rt_bvh_node*)dev.nodes, nn, o.data(), d.data(), o.size(), counts, &truncated);
            if (truncated) truncated_walks_add(truncated);
        }
        if (!on_device) (void)hipGetLastError();
    }
    if (!on_device) count_box_passes(tree.data(), nn, o.data(), d.data(), o.size(), counts, cancel);
    if (cancel.load()) return false;
    // the measured passes, plus a twentieth of their sum spread by surface area: boxes no probe ray met still fold sensibly
    std::vector<double> w(nn);
    double total = 0.0, area_sum = 0.0;
    {
        // (64 slices whatever the host, taken by up to 16 threads, their partial sums added in slice order: 4.9 M nodes on the headline scene, twice per adaptation)
        const unsigned K = nn >= 262144u ? 64u : 1u, T = K > 1u ? adapt_threads(nn, 131072) : 1u;
        std::vector<double> part_total(K, 0.0), part_area(K, 0.0);
        auto slice = [&](unsigned k, int pass, double prior)
        {
            double tt = 0.0, aa = 0.0;
            for (uint32_t n = (uint32_t)((uint64_t)nn * k / K), e = (uint32_t)((uint64_t)nn * (k + 1) / K); n < e; ++n)
            {
                if (pass == 1) { w[n] = (double)counts[n] + prior * w[n]; continue; }
                const rt_bvh_node& b = tree[n];
                const double dx = (double)b.bounds_max.x - b.bounds_min.x, dy = (double)b.bounds_max.y - b.bounds_min.y, dz = (double)b.bounds_max.z - b.bounds_min.z;
                w[n] = dx * dy + dy * dz + dz * dx;
                aa += w[n];
                tt += (double)counts[n];
            }
            if (pass == 0) { part_total[k] = tt; part_area[k] = aa; }
        };
        auto on_slices = [&](int pass, double prior)
        {
            std::atomic<unsigned> next{0};
            auto run = [&]() { for (unsigned k; (k = next.fetch_add(1)) < K;) slice(k, pass, prior); };
            std::vector<std::thread> pool;
            for (unsigned t = 1; t < T; ++t) pool.emplace_back(run);
            run();
            for (auto& th : pool) th.join();
        };
        on_slices(0, 0.0);
        for (unsigned k = 0; k < K; ++k) { total += part_total[k]; area_sum += part_area[k]; }
        if (!(total > 0.0) || !(area_sum > 0.0) || !std::isfinite(area_sum)) return false;
        on_slices(1, 0.05 * total / area_sum);
    }
    // what the fold on the device costs these rays: known before, and whether or not, a new fold can be built (a failed build
    // used to leave it 0, and a rotated candidate was then adopted without ever having been compared with it)
    cost[0] = cost[1] = 0.0;
    for (uint32_t r : roots_now) if (r < nn) cost[0] += w[r];
    cost[0] /= (double)o.size();
    std::vector<uint32_t> roots_new;
    bool folded = false;
    if (on_device)
    {
        WideNode* d_recs = nullptr;
        uint32_t n_recs = 0;
        folded = devfold::fold(dev.st, (const rt_bvh_node*)dev.nodes, nn, tree[0], nullptr, w.data(), &d_recs, &n_recs, &entry, &roots_new, &out, &cancel) && !out.empty();
        if (d_recs) (void)hipFree(d_recs);                                 // (the records travel with fold_upload, with the shadow side's slot order applied)
        (void)hipGetLastError();
        if (cancel.load()) return false;
    }
    if (!folded && (!build_wide_bvh(tree.data(), nn, RT_WIDE_SAH, out, entry, &roots_new, nullptr, w.data(), &cancel) || out.empty())) return false;
    for (uint32_t r : roots_new) cost[1] += w[r];
    cost[1] /= (double)o.size();
    if (pairs && roots_new.size() == out.size())
    {
        pair_layout_by_node_weights(out, roots_new, w.data(), nn);
    }
    if (roots_out) roots_out->swap(roots_new);
    return cost[1] < cost[0];
}

// The shadow rays' side of an adaptation.  Their verdict does not depend on the tree above the reference's leaves (own_bvh.h), so with mode bit 3
// the binary tree itself is first rotated for the probe rays' measured crossings (tree_rotate.h) and then folded; whichever of the two folds --
// of the tree as it was, of the rotated tree -- costs the probe rays less is the candidate.
bool adapt_shadow_candidate(FoldAdapt* a)
{
    const std::vector<rt_bvh_node>& tree = a->bvh2_sh.empty() ? a->bvh2 : a->bvh2_sh;
    const std::vector<uint32_t>& roots = a->roots_sh.empty() ? a->roots : a->roots_sh;
    a->rotations = 0;
    a->bvh2_sh_new.clear();
    const int fold_device = a->worker_fold_device();
    auto t_stage = std::chrono::steady_clock::now();
    auto stage = [&](int k) { const auto t = std::chrono::steady_clock::now(); a->stage_s[k] = std::chrono::duration<double>(t - t_stage).count(); t_stage = t; };
    // the fold of the tree as it is and the rotations do not need each other: they run side by side (with the folds on the device -- bit 1 -- and on host threads alike:
    // beside an orbiting camera's frames the adapted fold then lands after 0.56 - 0.62 s instead of 0.69 - 0.74 and the frames cost the same, profiles/r06/call26*.log)
    bool ok = false;
    const bool rotating = (a->mode.load() & 8u) && !a->sh_o.empty();
    std::thread plain_thread;
    auto plain_fold = [&, t0 = t_stage]()
    {
        ok = refold_for_rays(tree, a->sh_o, a->sh_d, roots, a->wide_sh, a->entry_sh, a->cost[1], a->cancel, &a->roots_sh_new, fold_device, a->pairs);
        a->stage_s[2] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    };
    if (rotating) plain_thread = std::thread(plain_fold); else plain_fold();
    if (!rotating || a->cancel.load()) { if (plain_thread.joinable()) plain_thread.join(); return ok; }
    std::vector<rt_bvh_node> rotated;
    double crossings[2] = {0.0, 0.0};
    const uint32_t made = treerot::rotate(tree.data(), (uint32_t)tree.size(), (const float*)a->sh_o.data(), (const float*)a->sh_d.data(), a->sh_o.size(), 8, rotated, crossings, &a->cancel, 3, 0.03, a->rotate_s);
    stage(3);
    if (plain_thread.joinable()) plain_thread.join();
    if (made == 0 || rotated.size() != tree.size() || a->cancel.load()) return ok;
    std::vector<WideNode> wide;
    std::vector<uint32_t> roots_rot;
    uint32_t entry = 0;
    double cost[2] = {0.0, 0.0};
    const std::vector<uint32_t> top{0u};                                   // (the rotated tree has no current fold: only cost[1] is read)
    (void)refold_for_rays(rotated, a->sh_o, a->sh_d, top, wide, entry, cost, a->cancel, &roots_rot, fold_device, a->pairs);
    stage(4);
    if (wide.empty() || roots_rot.empty() || a->cancel.load()) return ok;
    // the rotated tree's boxes differ, so its measured passes are compared as they are (both are box passes per probe ray at record roots);
    // without a known cost of the fold on the device nothing is adopted
    if (!(a->cost[1][0] > 0.0)) return ok;
    const double current = a->cost[1][0], plain = ok ? a->cost[1][1] : current;
    if (!(cost[1] < plain)) return ok;
    a->wide_sh.swap(wide); a->entry_sh = entry; a->roots_sh_new.swap(roots_rot); a->bvh2_sh_new.swap(rotated);
    a->cost[1][1] = cost[1];
    a->rotations = made;
    return cost[1] < current;
}

bool adapt_shadow_side(FoldAdapt* a)
{
    a->reordered = 0;
    // mode bit 4 needs every probe shadow ray's nearest occluder: a walk of the reference's tree that depends on neither the rotations nor the folds, so -- when the
    // folds are the device's and the host's threads are this worker's (bit 1) -- it runs beside them
    const bool want_order = (a->mode.load() & 16u) && !a->tri9.empty() && !a->sh_o.empty();
    std::vector<uint32_t> prim;
    double t_occ = 0.0;
    auto find_occluders = [&]()
    {
        const auto t0 = std::chrono::steady_clock::now();
        nearest_occluders(a->bvh2, a->tri9, a->sh_o, a->sh_d, prim, a->cancel);
        t_occ = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    };
    std::thread beside;
    if (want_order && a->worker_fold_device() >= 0) beside = std::thread(find_occluders);
    const bool ok = adapt_shadow_candidate(a);
    if (beside.joinable()) beside.join();
    if (ok && want_order && !a->wide_sh.empty() && !a->cancel.load())
    {
        // the candidate's slots, likeliest occluder first
        const auto t0 = std::chrono::steady_clock::now();
        if (prim.size() != a->sh_o.size()) find_occluders();
        const std::vector<rt_bvh_node>& tree = a->rotations != 0 ? a->bvh2_sh_new : (a->bvh2_sh.empty() ? a->bvh2 : a->bvh2_sh);
        if (!a->cancel.load()) a->reordered = occluder_first(a->wide_sh, a->roots_sh_new, tree, prim);
        a->stage_s[5] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    (void)t_occ;
    return ok;
}

// The probe's queues, as the asynchronous copies left them in the pinned staging area, become the rays the folds are made for
// (closest-hit rays clipped at their hit: a ray that hit something never visits what lies behind the hit).
void probe_unpack(FoldAdapt* a)
{
    if (!a->staging || a->probe_paths == 0) return;                    // rays given directly (rt_debug_fold_abandon)
    a->o.clear(); a->d.clear(); a->sh_o.clear(); a->sh_d.clear();
    for (uint32_t sample = 0; sample < a->probe_samples; ++sample)
    {
        DCounters h;
        memcpy(&h, a->staging + a->probe_counters(sample), sizeof(h));
        for (uint32_t bounce = 0; bounce < a->probe_bounces; ++bounce)
        {
            const uint32_t n = h.queue[bounce], ns = h.shadow[bounce];
            if (n > a->probe_paths || ns > a->probe_paths) { a->o.clear(); a->d.clear(); a->sh_o.clear(); a->sh_d.clear(); return; }
            if (n == 0) break;
            const float4* o = (const float4*)(a->staging + a->probe_block(sample, bounce, 0));
            const float4* d = (const float4*)(a->staging + a->probe_block(sample, bounce, 1));
            const float4* hits = (const float4*)(a->staging + a->probe_block(sample, bounce, 2));
            const size_t at = a->o.size();
            a->o.insert(a->o.end(), o, o + n);
            a->d.insert(a->d.end(), d, d + n);
            for (uint32_t i = 0; i < n; ++i)
            {
                uint32_t prim;
                memcpy(&prim, &hits[i].z, 4);
                if (prim != RT_INVALID_ID && hits[i].w > 0.0f && hits[i].w * 1.0001f < a->o[at + i].w) a->o[at + i].w = hits[i].w * 1.0001f;
            }
            if (ns != 0)
            {
                const float4* so = (const float4*)(a->staging + a->probe_block(sample, bounce, 3));
                const float4* sd = (const float4*)(a->staging + a->probe_block(sample, bounce, 4));
                a->sh_o.insert(a->sh_o.end(), so, so + ns);
                a->sh_d.insert(a->sh_d.end(), sd, sd + ns);
            }
        }
    }
}

// The adapted records go to the device from HERE, on a stream of the worker's own; what earlier adoptions replaced is freed here too, after
// a device synchronisation that only this thread waits for.
void fold_upload(FoldAdapt* a)
{
    if (a->device < 0 || a->cancel.load() || !(a->ok || a->ok_sh)) return;
    a->upload_failed = true;
    if (hipSetDevice(a->device) != hipSuccess) return;
    if (!a->retired.empty())
    {
        if (hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); return; }
        for (void* p : a->retired) (void)hipFree(p);
        a->retired.clear();
    }
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return; }
    bool ok = true;
    if (a->ok) ok = hipMalloc(&a->new_cl, a->wide.size() * sizeof(WideNode)) == hipSuccess &&
                    hipMemcpyAsync(a->new_cl, a->wide.data(), a->wide.size() * sizeof(WideNode), hipMemcpyHostToDevice, st) == hipSuccess;
    if (ok && a->ok_sh) ok = hipMalloc(&a->new_sh, a->wide_sh.size() * sizeof(WideNode)) == hipSuccess &&
                             hipMemcpyAsync(a->new_sh, a->wide_sh.data(), a->wide_sh.size() * sizeof(WideNode), hipMemcpyHostToDevice, st) == hipSuccess;
    ok = hipStreamSynchronize(st) == hipSuccess && ok;
    (void)hipStreamDestroy(st);
    if (!ok)
    {
        (void)hipGetLastError();
        if (a->new_cl) (void)hipFree(a->new_cl);
        if (a->new_sh) (void)hipFree(a->new_sh);
        a->new_cl = a->new_sh = nullptr;
        return;
    }
    a->upload_failed = false;
}

void fold_adapt_worker(FoldAdapt* a)
{
    const auto t0 = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    for (double& t : a->stage_s) t = 0.0;
    probe_unpack(a);
    a->stage_s[0] = since();
    if (!a->o.empty())
    {
        std::thread shadow([a]() { a->ok_sh = adapt_shadow_side(a); });
        a->ok = refold_for_rays(a->bvh2, a->o, a->d, a->roots, a->wide, a->entry, a->cost[0], a->cancel, &a->roots_new, a->worker_fold_device(), a->pairs);
        a->stage_s[1] = since() - a->stage_s[0];
        shadow.join();
        const double t_up = since();
        fold_upload(a);
        a->stage_s[6] = since() - t_up;
    }
    a->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    a->finished.store(true);
}

FoldAdapt::~FoldAdapt()
{
    cancel.store(true);
    if (worker.joinable()) worker.join();
    if (device >= 0)
    {
        // (the callers -- rt_scene_upload, rt_ctx_destroy -- free the scene's own records the same way: after their stream synchronisation)
        (void)hipSetDevice(device);
        if (probe_done) { (void)hipEventSynchronize(probe_done); (void)hipEventDestroy(probe_done); }
        if (probe) (void)rt_frame_destroy(probe);
        if (staging) (void)hipHostFree(staging);
        for (void* p : {new_cl, new_sh}) if (p) (void)hipFree(p);
        if (!retired.empty()) (void)hipDeviceSynchronize();
        for (void* p : retired) (void)hipFree(p);
    }
}
