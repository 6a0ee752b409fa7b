// samples_ahead_impl.h -- part of rt_hip.hip's translation unit (included inside its extern "C" block, before rt_advance_sample): RT_OPT_SAMPLES_AHEAD.
#pragma once

// ---- RT_OPT_SAMPLES_AHEAD: samples traced ahead of the caller's Integrate() calls -------------------------------------------------------
// The reference's own pattern is one Integrate() per frame at one sample per pixel (src/render.cpp:197); while the camera stands still -- its
// progressive accumulation -- sample s + 1 .. s + k are known the moment sample s is: same camera, consecutive sample indices.  A launch of one
// sample per pixel is its own tail (3.3 ms per 1080p frame where the work is worth 1.6), a launch of k samples is not.  So,
// once `RT_AHEAD_QUIET` samples have been advanced without a reset, the frame traces BATCHES of the next samples into two banks (frames of its
// own: same tile, same options, the rt_integrate schedule without its replay) on streams beside the context's, 2, 4, 8 .. `depth` samples at a time,
// and a later rt_advance_sample whose sample sits in a bank only REPLAYS that sample's slot of the bank's radiance log into the radiance
// (k_flush, first_slot) -- the sum after every Integrate() is the reference's bit for bit, sample by sample, in sample order.  One bank is
// consumed while the other computes; the moment a bank is empty its next batch is enqueued, so the device always has one batch running and one
// queued.  A reset, another camera, another option, a scene upload, rt_integrate or anything that looks between two stages DISCARDS what was
// traced ahead (its launches finish on their own streams, unobserved): the price of a camera that starts to move is at most 2 x depth samples
// of device time, once; a camera that moves every frame never leaves the quiet phase and pays nothing.
#define RT_AHEAD_QUIET 3u

static bool ahead_bank_idle(const AheadBank& b) { return b.next >= b.n; }

// the bank that holds sample `s` as its next slot, or -1
static int ahead_holds(const rt_frame* f, uint32_t s)
{
    if (!f->ahead || !f->ahead_opt) return -1;
    const Ahead& A = *f->ahead;
    if (A.scene != f->ctx->scene_uploads || memcmp(&A.camera, &f->camera, sizeof(rt_camera)) != 0) return -1;
    for (int i = 0; i < 2; ++i)
        if (A.bank[i].h && !ahead_bank_idle(A.bank[i]) && A.bank[i].base + A.bank[i].next == s) return i;
    return -1;
}

// Samples per batch: the caller's (2 .. 64), or -- 1 = automatic -- what makes a batch ~32 M paths: 16 samples of a 1080p frame, 4 of a 4K one
// (rt_integrate at 2 / 4 / 8 / 16 samples of the 1080p headline frame in flight: 2.67 / 2.26 / 1.97 / 1.80 ms per sample where one alone costs 3.3 and
// 128 together 1.56; the 4K / 16-bounce config: 22.7 / 19.4 / 17.9 ms at 1 / 2 / 4 -- profiles/r06_call01.log), within 64 GiB of path state for the two banks.
static uint32_t ahead_depth(const rt_frame* f)
{
    const uint64_t n = f->n_local ? f->n_local : 1u;
    uint64_t k = f->ahead_opt & 0xFFu;
    if (k == 0) return 0;
    if (k == 1 || k == 255) { k = (32000000ull + n - 1) / n; if (k > 64) k = 64; }
    const uint64_t budget = f->state_limit_mb ? ((uint64_t)f->state_limit_mb << 20) : (64ull << 30);
    const uint64_t per_sample = 2ull * n * (11u * 16u + 5u * 4u + 12u * 2u * (f->max_bounces + 1u));
    if (k * per_sample > budget) k = budget / per_sample;
    return k >= 2 ? (uint32_t)k : 0u;
}

// Is the stage API's next sample one this mode may serve?  (One sample of the whole tile in one chunk on the context's stream, nothing that reads
// between the stages.)
static bool ahead_wanted(const rt_frame* f)
{
    return f->ahead_opt != 0u && !f->ahead_owner && f->n_local != 0u && !(f->denoiser || f->aov != 0) && !f->profile && !f->timeline &&
           f->stage_pipes <= 1u && f->pipelines == 1u && f->ctx->scene.valid && ahead_depth(f) >= 2u;
}

static void ahead_mirror(const rt_frame* f, uint32_t (&m)[16])
{
    const uint32_t v[16] = {f->max_bounces, f->sampler, f->white_furnace, f->drop_last, f->overlap_shadow, f->trace_variant, f->trace_tune, f->shade_partition,
        f->trace_tail_lanes, f->chunk_refill, f->trace_waves_per_cu, f->select_form_box ? 1u : 0u, f->small_launch_set ? (uint32_t)std::min<uint64_t>(f->small_launch_paths, 0xFFFFFFFFull) : 0xFFFFFFFFu,
        (uint32_t)std::min<uint64_t>(f->trace_tail_paths, 0xFFFFFFFFull), ahead_depth(f), f->ahead_opt & 0x100u};
    memcpy(m, v, sizeof(v));
}

// The banks exist, are laid out for `depth` samples in flight and have the owner's options.  (Anything here may wait for the device: it runs when
// the mode starts and after an option has changed, never between two frames of a quiet camera.)
static int ahead_configure(rt_frame* f)
{
    rt_ctx* ctx = f->ctx;
    if (!f->ahead) f->ahead = new Ahead();
    Ahead& A = *f->ahead;
    uint32_t want[16];
    ahead_mirror(f, want);
    if (A.configured && memcmp(want, A.mirrored, sizeof(want)) == 0) return RT_OK;
    ahead_discard(f);
    A.configured = false;
    const bool two_streams = (f->ahead_opt & 0x100u) != 0u;
    for (int i = 0; i < 2; ++i)
    {
        // the banks' launches go beside the frame's own: one stream for both banks (their batches in order) or one each (they overlap)
        if (!A.stream[i] && (i == 0 || two_streams)) HIPCHK(ctx, hipStreamCreateWithFlags(&A.stream[i], hipStreamNonBlocking));
        AheadBank& b = A.bank[i];
        hipStream_t const st = two_streams ? A.stream[i] : A.stream[0];
        if (b.h && b.h->ps[0].stream != st) { (void)rt_frame_destroy(b.h); b.h = nullptr; }
        if (!b.h)
        {
            rt_frame_desc fd = {f->tile.width, f->tile.height, f->tile.rank, f->tile.nranks, f->tile.band_h};
            if (create_frame(ctx, &fd, &b.h, st) != RT_OK) { b.h = nullptr; return RT_ERROR; }
            b.h->ahead_owner = f;
        }
        if (!b.done) HIPCHK(ctx, hipEventCreateWithFlags(&b.done, hipEventDisableTiming));
        if (!b.order) HIPCHK(ctx, hipEventCreateWithFlags(&b.order, hipEventDisableTiming));
        if (sync_frame_streams(b.h) != RT_OK) return RT_ERROR;
        rt_frame* h = b.h;
        const std::pair<int, uint32_t> options[] = {{RT_OPT_MAX_BOUNCES, f->max_bounces}, {RT_OPT_SAMPLER, f->sampler}, {RT_OPT_WHITE_FURNACE, f->white_furnace},
            {RT_OPT_TRACE_DROP_LAST_BOUNCE_RAYS, f->drop_last}, {RT_OPT_OVERLAP_SHADOW, f->overlap_shadow}, {RT_OPT_TRACE_VARIANT, f->trace_variant},
            {RT_OPT_TRACE_TUNE, f->trace_tune}, {RT_OPT_SHADE_PARTITION, f->shade_partition}, {RT_OPT_TRACE_TAIL_LANES, f->trace_tail_lanes},
            {RT_OPT_CHUNK_REFILL, f->chunk_refill}, {RT_OPT_TRACE_WAVES_PER_CU, f->trace_waves_per_cu}, {RT_OPT_TRACE_SELECT_FORM_BOX, f->select_form_box ? 1u : 0u}};
        for (const auto& o : options)
            if (rt_set_option(h, o.first, o.second) != RT_OK) return RT_ERROR;
        h->trace_tail_paths = f->trace_tail_paths;
        h->small_launch_paths = f->small_launch_paths; h->small_launch_set = f->small_launch_set;
        if (ensure_slots(h, want[14]) != RT_OK) return RT_ERROR;
        if (h->slots < 2u || h->chunk_pixels < (f->n_local ? f->n_local : 1u)) return fail(ctx, "RT_OPT_SAMPLES_AHEAD: a bank could not be laid out for the whole tile");
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));       // the banks' allocations were cleared on the context's stream
    A.depth = std::min(want[14], std::min(A.bank[0].h->slots, A.bank[1].h->slots));
    memcpy(A.mirrored, want, sizeof(want));
    A.configured = true;
    return RT_OK;
}

// A batch: samples base .. base + n - 1 through the wavefront loop of bank i, as rt_integrate runs one (the same launches in the same order), but the
// log is left as it is: its replay happens sample by sample, by ahead_consume.
static int ahead_launch(rt_frame* f, int i, uint32_t base, uint32_t n)
{
    rt_ctx* ctx = f->ctx;
    Ahead& A = *f->ahead;
    AheadBank& b = A.bank[i];
    rt_frame* h = b.h;
    hipStream_t const st = h->ps[0].stream;
    h->camera = f->camera;
    h->sample_count = base;
    // behind whatever the owner's stream holds: the last replay out of this bank's log
    HIPCHK(ctx, hipEventRecord(b.order, ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(st, b.order, 0));
    h->p = &h->ps[0];
    h->fused = true;
    h->side_active = side_on(h);
    int rc = generate_rays(h, n, 0, false);
    if (rc == RT_OK) rc = rt_intersect(h, 0);
    for (uint32_t bounce = 0; bounce <= h->max_bounces && rc == RT_OK; ++bounce)
    {
        if (rt_shade(h, bounce) != RT_OK) rc = RT_ERROR;
        else if (h->side_active && bounce < h->max_bounces && rt_intersect(h, bounce + 1u) != RT_OK) rc = RT_ERROR;
        else if (rt_intersect_shadow(h, bounce) != RT_OK) rc = RT_ERROR;
        else if (!h->side_active && bounce < h->max_bounces && rt_intersect(h, bounce + 1u) != RT_OK) rc = RT_ERROR;
    }
    if (rc == RT_OK && (wait_shadow(h, 0) != RT_OK || wait_shadow(h, 1) != RT_OK)) rc = RT_ERROR;
    h->fused = false;
    if (rc == RT_OK && hipEventRecord(b.done, st) != hipSuccess) rc = fail(ctx, "RT_OPT_SAMPLES_AHEAD: recording a batch's end failed");
    if (rc != RT_OK)
    {
        // nothing of a batch that could not be enqueued is ever replayed
        (void)hipGetLastError();
        (void)sync_frame_streams(h);
        (void)hipMemsetAsync(h->ps[0].cnt, 0, (size_t)h->log_stride * sizeof(uint32_t), st);
        h->ps[0].cur_slots = 0; h->ps[0].shadow_pending = false; h->ps[0].shadow_in_flight[0] = h->ps[0].shadow_in_flight[1] = false;
        b.n = b.next = 0;
        return RT_ERROR;
    }
    b.base = base; b.n = n; b.next = 0;
    A.camera = f->camera;
    A.scene = ctx->scene_uploads;
    A.last_n = n;
    A.launched += n;
    return RT_OK;
}

// rt_advance_sample for a sample that sits in bank i: its slot of the bank's log, replayed into the radiance on the context's stream
static int ahead_consume(rt_frame* f, int i)
{
    rt_ctx* ctx = f->ctx;
    AheadBank& b = f->ahead->bank[i];
    rt_frame* h = b.h;
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, b.done, 0));
    const uint32_t blocks = (f->n_local + 255u) / 256u;
    hipLaunchKernelGGL(k_flush, dim3(blocks), dim3(256), 0, ctx->stream, f->radiance, dlog(h), f->n_local, 1u, h->chunk_pixels, 0u, b.next);
    HIPCHK(ctx, hipGetLastError());
    ++b.next;
    ++f->sample_count;
    ++f->ahead->consumed;
    if (ahead_bank_idle(b)) h->ps[0].cur_slots = 0;       // every slot replayed (k_flush has set their counts back to zero)
    return RT_OK;
}

// Every idle bank gets the next batch -- the samples behind the last one traced ahead -- as long as some bank holds the next sample (otherwise the
// mode is starting: `first` = the sample to begin with).  A failure to launch only ends the speculation: the samples are traced when they are asked for.
static void ahead_schedule(rt_frame* f, bool starting)
{
    if (!ahead_wanted(f)) return;
    if (ahead_configure(f) != RT_OK) { (void)hipGetLastError(); f->ahead_opt = 0; return; }   // (e.g. no memory for the banks: the mode switches itself off)
    Ahead& A = *f->ahead;
    if (A.depth < 2u) return;
    const uint32_t s = f->sample_count;
    if (!starting && ahead_holds(f, s) < 0 && !(ahead_bank_idle(A.bank[0]) && ahead_bank_idle(A.bank[1]))) return;
    for (int i = 0; i < 2; ++i)
    {
        if (!ahead_bank_idle(A.bank[i])) continue;
        uint32_t end = s;
        for (const AheadBank& o : A.bank) if (!ahead_bank_idle(o)) end = std::max(end, o.base + o.n);
        const uint32_t n = ahead_bank_idle(A.bank[i ^ 1]) ? 2u : std::min(A.depth, 2u * std::max(1u, A.last_n));
        if (end > 0xFFFFFFFFu - n) return;
        if (ahead_launch(f, i, end, n) != RT_OK) { (void)hipGetLastError(); return; }
        if (starting) return;                               // the first batch alone: the ramp's next step follows at its first replay
    }
}

static void ahead_discard(rt_frame* f)
{
    if (!f || !f->ahead) return;
    Ahead& A = *f->ahead;
    A.quiet = 0;
    A.last_n = 0;
    for (AheadBank& b : A.bank)
    {
        if (!b.h) continue;
        rt_frame* h = b.h;
        hipStream_t const st = h->ps[0].stream;
        if (!ahead_bank_idle(b) || h->ps[0].cur_slots != 0)
        {
            // behind the owner's last replay out of this log AND behind the batch itself (same stream): the slots nobody replayed go back to zero, and so
            // do the bank's ray counters (rt_frame_get_stats adds them to the owner's)
            (void)hipEventRecord(b.order, f->ctx->stream);
            (void)hipStreamWaitEvent(st, b.order, 0);
            (void)hipMemsetAsync(h->ps[0].cnt, 0, (size_t)h->log_stride * sizeof(uint32_t), st);
            A.discarded += b.n - b.next;
        }
        if (h->ps[0].counters) (void)hipMemsetAsync(h->ps[0].counters, 0, sizeof(DCounters), st);
        h->ps[0].cur_slots = 0; h->ps[0].shadow_pending = false; h->ps[0].prev_bounces = 0; h->ps[0].fold_accumulates = 0;
        b.n = b.next = 0;
    }
}

static void ahead_destroy(rt_frame* f)
{
    if (!f || !f->ahead) return;
    Ahead* A = f->ahead;
    f->ahead = nullptr;
    for (AheadBank& b : A->bank)
    {
        if (b.h) (void)rt_frame_destroy(b.h);             // (waits for its streams)
        if (b.done) (void)hipEventDestroy(b.done);
        if (b.order) (void)hipEventDestroy(b.order);
    }
    for (hipStream_t st : A->stream) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    delete A;
}

// A sample the frame traced itself has been advanced: RT_AHEAD_QUIET of them in a row (no reset between) start the mode -- a first batch of two.
static void ahead_after_plain_sample(rt_frame* f)
{
    if (!f->ahead_opt || !ahead_wanted(f)) return;
    if (!f->ahead) f->ahead = new Ahead();
    Ahead& A = *f->ahead;
    if (++A.quiet < RT_AHEAD_QUIET) return;
    if (ahead_bank_idle(A.bank[0]) && ahead_bank_idle(A.bank[1])) ahead_schedule(f, true);
}

