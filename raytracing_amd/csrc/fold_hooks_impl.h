// fold_hooks_impl.h -- part of rt_hip.hip's translation unit (included inside its extern "C" block, before rt_integrate): the render-thread side of
// RT_CTX_OPT_ADAPTIVE_FOLD -- the asynchronous probe frame, the hand-over to the worker (fold_adapt_impl.h), the adoption by pointer exchange, the trigger.
#pragma once

// ---- RT_CTX_OPT_ADAPTIVE_FOLD: probe, worker hand-over, adoption (FoldAdapt) -------------------------------------------------
// The probe: a frame of the same camera at 1/k of the resolution (about 32 K paths), one sample (several for tiny images), taken
// through the stage API.  Round 5: nothing here waits for the device -- every queue travels to pinned host memory with an asynchronous copy
// enqueued right behind the stage that filled it (whole capacity: the counters that say how much of it is rays come back last), an event
// marks the end, and the frame, the staging area and the event are kept for the scene's life (round 4: three blocking copies per bounce,
// frame created and destroyed per probe -- what an orbiting camera paid at every re-adaptation, VERDICT r04 / ADVICE r04).
static int fold_probe_enqueue(rt_frame* f, FoldAdapt& a)
{
    rt_ctx* ctx = f->ctx;
    const uint64_t pixels = (uint64_t)f->tile.width * f->tile.height;
    uint32_t k = 1;
    while (pixels / ((uint64_t)k * k) > 32768u) ++k;
    rt_frame_desc desc;
    desc.width = std::max(1u, f->tile.width / k); desc.height = std::max(1u, f->tile.height / k);
    desc.tile_rank = 0; desc.tile_count = 1; desc.band_height = desc.height;
    const uint32_t paths = desc.width * desc.height;
    const uint32_t n_samples = std::min(16u, std::max(1u, 32768u / std::max(1u, paths)));
    const uint32_t n_bounces = f->max_bounces + 1u;
    if (a.probe && (a.probe->tile.width != desc.width || a.probe->tile.height != desc.height)) { (void)rt_frame_destroy(a.probe); a.probe = nullptr; }
    if (!a.probe && rt_frame_create(ctx, &desc, &a.probe) != RT_OK) { a.probe = nullptr; return RT_ERROR; }
    rt_frame* p = a.probe;
    if (!a.probe_done && hipEventCreateWithFlags(&a.probe_done, hipEventDisableTiming) != hipSuccess) { a.probe_done = nullptr; return fail(ctx, "rt_integrate: the probe frame's event could not be created"); }
    a.probe_paths = paths; a.probe_samples = n_samples; a.probe_bounces = n_bounces;
    const size_t need = a.probe_counters(n_samples);
    if (need > a.staging_bytes)
    {
        if (a.staging) (void)hipHostFree(a.staging);
        a.staging = nullptr; a.staging_bytes = 0;
        if (hipHostMalloc((void**)&a.staging, need, hipHostMallocDefault) != hipSuccess) { a.staging = nullptr; (void)hipGetLastError(); return fail(ctx, "rt_integrate: no pinned memory for the probe frame's queues"); }
        a.staging_bytes = need;
    }
    int rc = rt_reset(p);                                                // sample 0 again, like the fresh frame of round 4's probe
    const std::pair<int, uint32_t> options[] = {{RT_OPT_MAX_BOUNCES, f->max_bounces}, {RT_OPT_SAMPLER, f->sampler}, {RT_OPT_WHITE_FURNACE, f->white_furnace},
        {RT_OPT_TRACE_DROP_LAST_BOUNCE_RAYS, f->drop_last}, {RT_OPT_OVERLAP_SHADOW, 0u}};
    for (const auto& o : options)
        if (rc == RT_OK && rt_set_option(p, o.first, o.second) != RT_OK) rc = RT_ERROR;
    if (rc == RT_OK && rt_set_camera(p, &f->camera) != RT_OK) rc = RT_ERROR;
    if (rc == RT_OK && (p->log_stride < paths || p->chunk_pixels < paths)) rc = fail(ctx, "rt_integrate: the probe frame's queues are smaller than its image");
    auto back = [&](size_t at, const void* src, size_t bytes) -> bool
    {
        return hipMemcpyAsync(a.staging + at, src, bytes, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
    };
    const size_t q = (size_t)paths * sizeof(float4);
    for (uint32_t sample = 0; sample < n_samples && rc == RT_OK; ++sample)
    {
        if (rt_generate_rays(p) != RT_OK) { rc = RT_ERROR; break; }
        for (uint32_t bounce = 0; bounce < n_bounces && rc == RT_OK; ++bounce)
        {
            const uint32_t in = bounce & 1u;
            if (rt_intersect(p, bounce) != RT_OK) { rc = RT_ERROR; break; }
            if (!back(a.probe_block(sample, bounce, 0), p->p->o4[in], q) || !back(a.probe_block(sample, bounce, 1), p->p->d4[in], q) ||
                !back(a.probe_block(sample, bounce, 2), p->p->hits, q)) { rc = RT_ERROR; break; }
            if (rt_shade(p, bounce) != RT_OK) { rc = RT_ERROR; break; }
            if (!back(a.probe_block(sample, bounce, 3), p->p->sh_o4[in], q) || !back(a.probe_block(sample, bounce, 4), p->p->sh_d4[in], q)) { rc = RT_ERROR; break; }
            if (rt_intersect_shadow(p, bounce) != RT_OK) rc = RT_ERROR;
        }
        // the sample's counters: queue[b] and shadow[b] of every bounce are still there (k_raygen resets them for the NEXT sequence)
        if (rc == RT_OK && !back(a.probe_counters(sample), p->p->counters, sizeof(DCounters))) rc = RT_ERROR;
        if (rc == RT_OK && rt_advance_sample(p) != RT_OK) rc = RT_ERROR;
    }
    if (rc == RT_OK && hipEventRecord(a.probe_done, ctx->stream) != hipSuccess) rc = RT_ERROR;
    if (rc != RT_OK)
    {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(ctx->stream);                         // whatever was enqueued writes the staging area: let it finish
    }
    return rc;
}

// The adapted folds replace the records the kernels are given from now on: an exchange of pointers (the worker has uploaded the new records;
// launches already enqueued keep reading the old ones, which the NEXT worker frees after a device synchronisation of its own thread).
static int fold_adopt(rt_ctx* ctx)
{
    Scene& s = ctx->scene;
    FoldAdapt* a = s.adapt;
    if (a->worker.joinable()) a->worker.join();
    a->state = FoldAdapt::IDLE;
    a->finished.store(false);
    char line[720];
    const size_t at = s.tree_report.find("adaptive fold");              // one line, the latest adaptation's
    if (at != std::string::npos) s.tree_report.erase(at);
    if (a->o.empty())
    {
        s.tree_report += "adaptive fold: the probe frame brought no rays back -> the fold stays as it is\n";
        a->state = FoldAdapt::OFF;
    }
    else if (a->upload_failed)
    {
        // the scene keeps the fold it has: a failed adaptation costs nothing but itself (and is not tried again)
        s.tree_report += "adaptive fold: not adopted (device allocation or copy failed)\n";
        a->state = FoldAdapt::OFF;
    }
    else
    {
        const bool shared = s.d.wnodes_sh == s.d.wnodes;               // the shadow rays walk the closest-hit records
        if (a->ok)
        {
            void* old = s.wnodes;
            s.wnodes = a->new_cl;
            s.d.wnodes = (const float4*)a->new_cl; s.d.w_entry_ref = a->entry; s.n_wide = (uint32_t)a->wide.size();
            if (shared && !a->ok_sh) { s.wnodes_sh = old; a->roots_sh = a->roots; s.n_wide_sh = (uint32_t)a->roots.size(); }   // ... and keep walking the old ones (theirs now)
            else if (old) a->retired.push_back(old);
            a->roots.swap(a->roots_new);
            a->new_cl = nullptr;
        }
        if (a->ok_sh)
        {
            if (s.wnodes_sh) a->retired.push_back(s.wnodes_sh);
            s.wnodes_sh = a->new_sh;
            s.d.wnodes_sh = (const float4*)a->new_sh; s.d.w_sh_entry_ref = a->entry_sh; s.n_wide_sh = (uint32_t)a->wide_sh.size();
            a->roots_sh.swap(a->roots_sh_new);
            if (a->rotations != 0) a->bvh2_sh.swap(a->bvh2_sh_new);          // the shadow rays' binary tree from now on
            a->new_sh = nullptr;
        }
        if (a->ok || a->ok_sh) ++a->adaptations;
        snprintf(line, sizeof(line), "adaptive fold (probe %u): %zu closest-hit and %zu shadow probe rays; box passes per probe ray at record roots: closest-hit %.2f -> %.2f (%s), "
            "shadow %.2f -> %.2f (%s); %.2f s on a worker thread (probe unpacked %.2f; closest-hit re-fold %.2f beside the shadow side's: plain re-fold %.2f, rotations %.2f = pointer form %.2f + ray lists %.2f + passes %.2f + linear layout %.2f, "
            "re-fold after them %.2f, occluder order %.2f; upload %.2f)\n", a->adaptations, a->o.size(), a->sh_o.size(), a->cost[0][0], a->cost[0][1], a->ok ? "adopted" : "kept",
            a->cost[1][0], a->cost[1][1], a->ok_sh ? "adopted" : "kept", a->seconds, a->stage_s[0], a->stage_s[1], a->stage_s[2], a->stage_s[3], a->rotate_s[0], a->rotate_s[1], a->rotate_s[2], a->rotate_s[3], a->stage_s[4], a->stage_s[5], a->stage_s[6]);
        s.tree_report += line;
        if (a->ok_sh && a->reordered != 0)
        {
            s.tree_report.pop_back();
            snprintf(line, sizeof(line), "; %u shadow records' slots stored likeliest occluder first\n", a->reordered);
            s.tree_report += line;
        }
        if (a->ok_sh && a->rotations != 0)
        {
            s.tree_report.pop_back();
            snprintf(line, sizeof(line), "; the shadow rays' binary tree rotated for the probe rays' crossings first (%u rotations)\n", a->rotations);
            s.tree_report += line;
        }
        const uint64_t truncated = truncated_walks_exchange();
        if (truncated != 0)
        {
            s.tree_report.pop_back();
            snprintf(line, sizeof(line), "; %llu host walks met a subtree deeper than their 126-entry stack (weights only)\n", (unsigned long long)truncated);
            s.tree_report += line;
        }
    }
    // the rays and the records have served; the binary trees stay for the next camera
    for (auto* v : {&a->o, &a->d, &a->sh_o, &a->sh_d}) std::vector<float4>().swap(*v);
    for (auto* v : {&a->wide, &a->wide_sh}) std::vector<WideNode>().swap(*v);
    for (auto* v : {&a->roots_new, &a->roots_sh_new}) std::vector<uint32_t>().swap(*v);
    std::vector<rt_bvh_node>().swap(a->bvh2_sh_new);
    return RT_OK;
}

// Has the camera left the view the folds were adapted to?  (tools/fold_weight_study.py --views: a fold adapted to one view costs another view
// 0 .. + 2 % against the surface-area fold as a rule and up to + 11 % -- street level seen with a fold made from above -- while its own view
// gains 2 .. 14 %.)  Position by 3 % of the scene's diagonal, direction by 20 degrees, field of view by a tenth.
static bool fold_view_left(const FoldAdapt& a, const rt_camera& c)
{
    const double dx = (double)c.position.x - a.camera.position.x, dy = (double)c.position.y - a.camera.position.y, dz = (double)c.position.z - a.camera.position.z;
    if (std::sqrt(dx * dx + dy * dy + dz * dz) > 0.03 * a.scene_diagonal) return true;
    const double la = std::sqrt((double)a.camera.front.x * a.camera.front.x + (double)a.camera.front.y * a.camera.front.y + (double)a.camera.front.z * a.camera.front.z);
    const double lc = std::sqrt((double)c.front.x * c.front.x + (double)c.front.y * c.front.y + (double)c.front.z * c.front.z);
    const double dot = (double)c.front.x * a.camera.front.x + (double)c.front.y * a.camera.front.y + (double)c.front.z * a.camera.front.z;
    if (la > 0.0 && lc > 0.0 && !(dot >= 0.9396926 * la * lc)) return true;
    return std::fabs((double)c.fov - a.camera.fov) > 0.1 * std::fabs((double)a.camera.fov);
}

static int fold_adapt_hook(rt_frame* f)
{
    Scene& s = f->ctx->scene;
    FoldAdapt* a = s.adapt;
    if (!a || a->state == FoldAdapt::OFF) return RT_OK;
    if (a->probe == f) return RT_OK;                                       // (the probe frame goes through the stage API, never through here)
    const bool eligible = !(f->denoiser || f->aov != 0 || f->n_local == 0);
    if (a->state == FoldAdapt::IDLE && eligible && fold_view_left(*a, f->camera))
    {
        // an orbiting camera leaves the view again and again: at most one adaptation per min_interval_ms (bit 1 -- tests, bench.py -- waits
        // for every one of them anyway)
        const auto now = std::chrono::steady_clock::now();
        if ((a->mode.load() & 2u) || std::chrono::duration<double, std::milli>(now - a->last_armed).count() >= (double)a->min_interval_ms.load()) a->state = FoldAdapt::ARMED;
    }
    if (a->state == FoldAdapt::ARMED)
    {
        if (!eligible) return RT_OK;                                       // another frame of this scene will do
        a->camera = f->camera;
        a->last_armed = std::chrono::steady_clock::now();
        if (fold_probe_enqueue(f, *a) != RT_OK)
        {
            a->state = FoldAdapt::OFF;
            const size_t at = s.tree_report.find("adaptive fold");
            if (at != std::string::npos) s.tree_report.erase(at);
            s.tree_report += "adaptive fold: the probe frame failed (" + f->ctx->error + ") -> the fold stays as it is\n";
            return RT_OK;
        }
        a->state = FoldAdapt::PROBING;
    }
    if (a->state == FoldAdapt::PROBING)
    {
        if (a->mode.load() & 2u) { if (hipEventSynchronize(a->probe_done) != hipSuccess) { (void)hipGetLastError(); a->state = FoldAdapt::OFF; return RT_OK; } }
        else
        {
            const hipError_t e = hipEventQuery(a->probe_done);
            if (e == hipErrorNotReady) return RT_OK;                       // the frame goes on with the fold it has
            if (e != hipSuccess) { (void)hipGetLastError(); a->state = FoldAdapt::OFF; return RT_OK; }
        }
        a->state = FoldAdapt::COMPUTING;
        a->ok = a->ok_sh = false;
        a->upload_failed = false;
        a->cost[0][0] = a->cost[0][1] = a->cost[1][0] = a->cost[1][1] = 0.0;
        a->finished.store(false);
        a->worker = std::thread(fold_adapt_worker, a);
    }
    if (a->state == FoldAdapt::COMPUTING && ((a->mode.load() & 2u) || a->finished.load())) return fold_adopt(f->ctx);
    return RT_OK;
}


// ---- one adaptation per process GROUP (round 6) ----------------------------------------------------------------------------------------
// N ranks that tile one image hold the same scene and probe the same low-resolution frame: N identical adaptations (tree rotations on host threads: seconds)
// for identical records.  Instead one rank adapts and the others TAKE its 4-wide records: rt_scene_export_folds copies the context's current records (the
// closest-hit rays' and the shadow rays', as adapted so far) to the host, the launcher's own channel carries them (torch.distributed, MPI, a file: 64 bytes
// per record, ~240 MB for the 2.8 M-triangle scene), rt_scene_import_folds puts them in place of the receiver's own and switches its adaptation off.  The
// records are indices into the triangle records every rank made from the same scene, so nothing else travels.  Exactness is the fold's: any fold of the same
// binary tree over the same leaves gives the same results -- and the importer checks what it can: record count, refs within range,
// leaf refs within the triangle array.
int rt_scene_export_folds(rt_ctx* ctx, void* closest_records, void* shadow_records, uint32_t capacity, uint32_t* n_closest, uint32_t* n_shadow, uint32_t* entries2)
{
    if (!ctx || !n_closest || !n_shadow) return fail(ctx, "rt_scene_export_folds: NULL argument");
    Scene& s = ctx->scene;
    if (!s.valid || !s.wide_ok || !s.wnodes) return fail(ctx, "rt_scene_export_folds: the scene has no 4-wide tree");
    (void)hipSetDevice(ctx->device);
    const bool own_shadow = s.d.wnodes_sh != s.d.wnodes && s.wnodes_sh != nullptr;
    const uint32_t n_cl = s.wnodes_cl != nullptr && s.d.wnodes == (const float4*)s.wnodes_cl ? s.n_wide_cl : s.n_wide;   // (tolerance mode: the closest-hit rays walk wnodes_cl)
    *n_closest = n_cl;
    *n_shadow = own_shadow ? s.n_wide_sh : 0u;
    if (entries2) { entries2[0] = s.d.w_entry_ref; entries2[1] = s.d.w_sh_entry_ref; }
    if (!closest_records) return RT_OK;                                                       // size query
    if (n_cl > capacity || *n_shadow > capacity) return fail(ctx, "rt_scene_export_folds: capacity too small");
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpy(closest_records, s.d.wnodes, (size_t)n_cl * sizeof(WideNode), hipMemcpyDeviceToHost));
    if (own_shadow && shadow_records) HIPCHK(ctx, hipMemcpy(shadow_records, s.d.wnodes_sh, (size_t)s.n_wide_sh * sizeof(WideNode), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_scene_import_folds(rt_ctx* ctx, const void* closest_records, uint32_t n_closest, uint32_t entry_closest, const void* shadow_records, uint32_t n_shadow, uint32_t entry_shadow)
{
    if (!ctx || !closest_records || n_closest == 0) return fail(ctx, "rt_scene_import_folds: NULL argument");
    Scene& s = ctx->scene;
    if (!s.valid || !s.wide_ok) return fail(ctx, "rt_scene_import_folds: the scene has no 4-wide tree to replace (upload it first, with RT_CTX_OPT_WIDE_BVH = 1)");
    if (n_closest >= (1u << 26) || n_shadow >= (1u << 26)) return fail(ctx, "rt_scene_import_folds: too many records");
    (void)hipSetDevice(ctx->device);
    // what can be checked: every ref is a record of the same array, a leaf inside the triangle array, or empty
    const uint32_t nt = s.n_tris;
    auto sane = [&](const WideNode* r, uint32_t n, uint32_t entry) -> bool
    {
        if (entry >= n) return false;
        for (uint32_t i = 0; i < n; ++i)
            for (uint32_t ref : r[i].ref)
            {
                if (ref == RT_EMPTY_REF) continue;
                if (ref & RT_LEAF_BIT) { if ((ref & ~RT_LEAF_BIT) >= nt) return false; }
                else if (ref >= n) return false;
            }
        return true;
    };
    if (!sane((const WideNode*)closest_records, n_closest, entry_closest)) return fail(ctx, "rt_scene_import_folds: the closest-hit records do not fit this scene (a ref outside the records / the triangles)");
    if (n_shadow && (!shadow_records || !sane((const WideNode*)shadow_records, n_shadow, entry_shadow))) return fail(ctx, "rt_scene_import_folds: the shadow records do not fit this scene");
    // nothing in flight may still read the records that go: batches traced ahead are dropped, every stream drains, an adaptation of this context's own is abandoned
    for (rt_frame* f : ctx->frames) ahead_discard(f);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (rt_frame* f : ctx->frames)
        if (sync_frame_streams(f) != RT_OK) return RT_ERROR;
    if (s.adapt) { drop_fold_adapt(s.adapt); s.adapt = nullptr; }
    void *cl = nullptr, *sh = nullptr;
    int rc = dev_alloc_copy(ctx, &cl, closest_records, (size_t)n_closest * sizeof(WideNode));
    if (rc == RT_OK && n_shadow) rc = dev_alloc_copy(ctx, &sh, shadow_records, (size_t)n_shadow * sizeof(WideNode));
    if (rc == RT_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(ctx, "rt_scene_import_folds: upload failed");
    if (rc != RT_OK) { if (cl) (void)hipFree(cl); if (sh) (void)hipFree(sh); (void)hipGetLastError(); return RT_ERROR; }
    const bool closest_is_own = s.wnodes_cl != nullptr && s.d.wnodes == (const float4*)s.wnodes_cl;      // (tolerance mode: the closest-hit rays walk wnodes_cl)
    if (closest_is_own) { (void)hipFree(s.wnodes_cl); s.wnodes_cl = cl; s.n_wide_cl = n_closest; }
    else { if (s.wnodes) (void)hipFree(s.wnodes); s.wnodes = cl; s.n_wide = n_closest; }
    s.d.wnodes = (const float4*)cl; s.d.w_entry_ref = entry_closest;
    if (s.wnodes_sh) { (void)hipFree(s.wnodes_sh); s.wnodes_sh = nullptr; s.n_wide_sh = 0; }
    if (n_shadow) { s.wnodes_sh = sh; s.n_wide_sh = n_shadow; s.d.wnodes_sh = (const float4*)sh; s.d.w_sh_entry_ref = entry_shadow; }
    else { s.d.wnodes_sh = s.d.wnodes; s.d.w_sh_entry_ref = s.d.w_entry_ref; }
    char line[200];
    const size_t at = s.tree_report.find("imported folds");
    if (at != std::string::npos) s.tree_report.erase(at);
    snprintf(line, sizeof(line), "imported folds: %u closest-hit + %u shadow records taken from another context of the group; this context's own adaptation is off\n", n_closest, n_shadow);
    s.tree_report += line;
    return RT_OK;
}
