// rt_hip.hip -- implementation of the C-ABI in include/rt_hip.h over HIP for
// gfx950.  Replaces the reference's src/gpu_wrappers/cl_context.cpp and the
// device-facing half of src/integrator/cl_pt_integrator.cpp.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <cmath>
#include <algorithm>
#include <string>
#include <vector>
#include <map>
#include <thread>
#include <atomic>
#include <array>
#include "rt_hip.h"
#include "kernels.h"
#include "own_bvh.h"
#include "treelet_order.h"
#include "tree_select.h"
#include "tree_rotate.h"
#include <chrono>
#include "wide_bvh.h"        // the host-side fold, the pair layout, the adaptation's host walks (wide_bvh.cpp)
#include "device_fold.h"     // ... and the fold + crossing counts on the device (device_fold.hip)
using namespace rtw;

namespace
{
thread_local std::string g_thread_error;

struct FoldAdapt;                          // RT_CTX_OPT_ADAPTIVE_FOLD: the state of a scene's fold adaptation (below, after choose_tree)
void drop_fold_adapt(FoldAdapt* a);        // waits for its worker thread
void fold_adapt_set_interval(FoldAdapt* a, uint32_t ms);
void fold_adapt_set_wait(FoldAdapt* a, bool wait);

struct Scene
{
    void* nodes = nullptr; void* tris_rt = nullptr; void* tris_sh = nullptr; void* materials = nullptr;
    void* textures = nullptr; void* texture_data = nullptr; void* lights = nullptr; void* env = nullptr;
    void* emissive = nullptr;
    void* mat_tex16 = nullptr;
    void* wnodes = nullptr;   // 4-wide quantized BVH (build_wide_bvh); nullptr when the tree does not qualify
    void* wnodes_sh = nullptr;   // the shadow rays' own 4-wide tree (own_bvh.h over the reference's leaves); nullptr = they share wnodes
    void* wnodes_cl = nullptr;   // RT_CTX_OPT_CLOSEST_TREE = 1 (tolerance mode): the closest-hit rays' own tree
    uint32_t n_wide_sh = 0, n_wide_cl = 0;
    uint32_t n_tris = 0;         // triangles of the uploaded scene (rt_scene_import_folds checks leaf refs against it)
    std::string tree_report;     // what rt_scene_upload measured when it chose the trees (rt_scene_tree_report)
    FoldAdapt* adapt = nullptr;  // RT_CTX_OPT_ADAPTIVE_FOLD: armed at upload, run by the first rt_integrate (fold_adapt_hook)
    DScene d = {};
    bool valid = false;
    uint32_t n_wide = 0;      // wide nodes (0 with a leaf root)
    bool wide_ok = false;     // build_wide_bvh succeeded (k_trace_w4 usable)
    bool offsets32 = false;   // node and trace-triangle arrays below 4 GiB: k_trace2 addresses them with 32-bit byte offsets
    // a quarter or more of the shadow rays will have a non-finite 1/dir component (directional lights along a coordinate
    // axis, e.g. an overhead light (0, -1, 0)): k_trace_w4 would hand every one of them to its small follow-up launch,
    // so the automatic choice traces the shadow queue with k_trace2 (select-form slab test inline, full residency)
    bool slow_shadow = false;
};
} // namespace

struct rt_ctx
{
    int device = 0;
    hipStream_t stream = nullptr;
    hipDeviceProp_t prop;
    std::string error;
    Scene scene;
    uint32_t treelet_nodes = 7;   // RT_CTX_OPT_TREELET_NODES
    uint32_t build_wide = 1;      // RT_CTX_OPT_WIDE_BVH
    uint32_t shadow_tree = 1;     // RT_CTX_OPT_SHADOW_TREE: 1 = shadow rays walk the backend's own tree where it measures cheaper (exact either way),
                                  // 2 = own unconditionally, 3 = own with the surface-area metric (A/B), 0 = they share the closest-hit tree
    uint32_t closest_tree = 0;    // RT_CTX_OPT_CLOSEST_TREE: 1 / 2 as above; != 0 is the tolerance mode (NOT bit-exact)
    uint32_t adaptive_fold = 25;  // RT_CTX_OPT_ADAPTIVE_FOLD (default bits 0 + 3 + 4 since round 5): bit 0 = re-fold the 4-wide trees for the rays rt_integrate actually traces (exact: a fold
                                  // decides which boxes are tested, never a result), bit 1 = rt_integrate waits for the new fold instead of
                                  // adopting it when it is ready, bit 2 = also for scenes too small to profit (tests), bit 3 = the shadow rays'
                                  // binary tree is rotated for the probe rays' crossings before it is folded (tree_rotate.h), bit 4 = the slots of
                                  // every shadow record are stored likeliest occluder first (measured on the device in round 5, profiles/r05_call01_*:
                                  // shadow trace 0.314 -> 0.258 ms per sample on the headline scene, bit-identical on all five configs)
    uint32_t adapt_min_interval_ms = 500;   // RT_CTX_OPT_ADAPT_MIN_INTERVAL_MS
    uint32_t wide_layout = 0;               // RT_CTX_OPT_WIDE_LAYOUT: 1 = the 4-wide records stored in (parent, likeliest child) pairs, one pair per 128-byte line (pair_layout)
    uint32_t tree_builder = 2;              // RT_CTX_OPT_TREE_BUILDER: the shadow rays' own binary tree -- 0 = own_bvh.h's full-sweep SAH on host threads, 1 = PLOC on the device (ploc_kernels.h),
                                            // 2 (default) = both start, the device's is measured first and the host's build is abandoned if it wins its measurement
    uint32_t device_fold = 1;               // RT_CTX_OPT_DEVICE_FOLD: the SAH collapse into 4-wide records runs on the device (fold_kernels.h); 0 = on host threads
    uint64_t scene_uploads = 0;             // rt_scene_upload calls so far (what a frame's measured choices were made for)
    std::vector<rt_frame*> frames;   // the frames alive on this context (rt_finish waits for their side streams too)
    uint8_t* blue_noise = nullptr;   // sobol[65536] | scramblingTile[131072] | rankingTile[131072]
    float* gamma_lut = nullptr;      // pow(byte / 255, 2.2f), 256 entries (k_fill_gamma_lut)
};

struct rt_buffer
{
    rt_ctx* ctx;
    void* ptr;
    size_t bytes;
};

#define RT_MAX_PIPES 4
// Per-path state of one chunk in flight and the stream its launches go to (see rt_frame::ps).
struct PathPipe
{
    hipStream_t stream = nullptr;      // pipe 0: the context's stream; others: their own
    hipEvent_t done = nullptr;         // cross-stream ordering (fork_pipes / join_pipes)
    // ray queues (ping-pong), hits, shadow queue
    float4* o4[2] = {nullptr, nullptr}; float4* d4[2] = {nullptr, nullptr};
    float4* thr[2] = {nullptr, nullptr};
    float4* hits = nullptr;
    // shadow queue, two of them: bounce b fills [b & 1] while the shadow trace of bounce b - 1 may still read the other
    float4* sh_o4[2] = {nullptr, nullptr}; float4* sh_d4[2] = {nullptr, nullptr}; uint32_t* sh_aux[2] = {nullptr, nullptr};
    // the shadow trace's own stream (rt_integrate: it runs beside the next bounce's closest-hit trace and k_shade,
    // filling the tail of one and the ramp of the other), its spill area and slow-ray list
    hipStream_t side = nullptr;
    hipEvent_t ev_shaded = nullptr, ev_shadow[2] = {nullptr, nullptr};
    bool shadow_in_flight[2] = {false, false};   // ev_shadow[i] recorded and not yet waited for by the main stream
    uint2* sh_spill = nullptr;
    uint32_t* sh_slow_list = nullptr;
    // radiance log (kernels_common.h header): cnt[id], rlog[entry][id]; id < slots * chunk_pixels
    float* rlog = nullptr; uint32_t* cnt = nullptr;   // rlog: 3 floats per entry
    uint32_t* ovf_slot = nullptr;      // compact log layout: a path's overflow block (DLog, kernels_common.h)
    uint32_t* slow_list = nullptr;     // queue indices k_trace_w4 leaves to k_trace2 (one per path)
    DCounters* counters = nullptr;
    uint2* spill = nullptr;
    uint32_t cur_slots = 0;            // slots used by the batch in flight on this pipe (0 = nothing pending)
    uint32_t chunk_base = 0;           // first local pixel of the chunk in flight
    uint32_t chunk_count = 0;          // pixels of the chunk in flight
    uint32_t prev_bounces = 0;
    uint32_t fold_accumulates = 0;     // the sequence in flight is a 2nd+ chunk of its batch: its counters ADD to last_*
    bool shadow_pending = false;       // rt_shade issued, rt_intersect_shadow not yet
};

// RT_OPT_SAMPLES_AHEAD: samples of the stage API traced ahead of the caller's Integrate() calls (below, "samples ahead").  A BANK is a frame of its
// own -- same tile, same options, its own per-path buffers -- that holds one batch; two banks, so that one computes while the other is consumed.
struct AheadBank
{
    rt_frame* h = nullptr;
    uint32_t base = 0, n = 0, next = 0;     // holds samples base .. base + n - 1; slots < next have reached the owner's radiance
    hipEvent_t done = nullptr;              // the batch's last launch (the bank's stream, after its side stream has been joined)
    hipEvent_t order = nullptr;             // what the bank's stream waits for before it starts another batch: the owner's last replay of the old one
};
struct Ahead
{
    AheadBank bank[2];
    hipStream_t stream[2] = {nullptr, nullptr};
    uint32_t quiet = 0;                     // samples advanced since the last reset / discard
    uint32_t depth = 0, last_n = 0;         // the most samples a batch holds (resolved for this tile); the latest batch's size (the ramp: 2, 4, .. depth)
    bool configured = false;
    uint32_t mirrored[16] = {};             // the owner's options as the banks have them
    rt_camera camera;                       // what the batches in flight were traced for
    uint64_t scene = 0;
    uint64_t launched = 0, consumed = 0, discarded = 0;   // samples
};

struct rt_frame
{
    rt_ctx* ctx;
    DTile tile;
    uint32_t n_local;
    float4* radiance; float4* resolved;
    // rt_frame_present: the resolved image goes to the host on a stream of its own while the next frame is traced
    float4* resolved_b = nullptr;      // the second device image (double buffering)
    hipStream_t present_stream = nullptr;
    hipEvent_t ev_resolved[2] = {nullptr, nullptr}, ev_copied[2] = {nullptr, nullptr};
    uint32_t present_flip = 0;
    bool present_pending = false;
    // Per-path state lives in PIPES (PathPipe): the tile is cut into chunks of pixels and chunk c travels through
    // the wavefront loop on pipe c % n_pipes, each pipe on its own HIP stream.  One chunk on one pipe is the plain
    // case.  More pipes let chunks overlap; measured on MI355X this does NOT pay (a trace launch costs ~0.8 ms
    // beyond its rays whatever shares the machine with it: profiles/r02_pipelines_sweep.log), so the default is 1.
    PathPipe ps[RT_MAX_PIPES];
    PathPipe* p = &ps[0];          // the pipe the stage functions work on
    uint32_t pipelines = 1;        // RT_OPT_PIPELINES: pipes rt_integrate may use
    uint32_t stage_pipes = 1;      // RT_OPT_STAGE_PIPES: ONE sample per pixel in flight (the stage API, rt_integrate(f, 1): the reference's frame-by-frame
                                   // pattern) is cut into this many chunks, each on a pipe (stream) of its own: every launch of that pattern is
                                   // its own tail, and the chunks' tails overlap
    uint32_t stage_chunks = 1;     // pipes holding a chunk of the stage API's sample in flight (set by rt_generate_rays)
    // RT_OPT_FRAME_KERNEL: the stage API's sample as ONE launch (k_frame, frame_kernels.h).  The stage calls of a sample are only RECORDED
    // while they come in the canonical order; rt_advance_sample launches the kernel.  Anything that needs the state between two stages
    // (a debug reader, the radiance mid-sample, another order of calls) first replays the recorded stages with the stage kernels.
    uint32_t frame_kernel = 0;
    struct { bool active = false; uint32_t bounce = 0; int next = 0; bool ahead = false; } deferred;   // next: 0 = rt_intersect(bounce), 1 = rt_shade, 2 = rt_intersect_shadow;
                                                                                                      // ahead: the sample is in a bank (RT_OPT_SAMPLES_AHEAD)
    uint32_t ahead_opt = 0;            // RT_OPT_SAMPLES_AHEAD
    Ahead* ahead = nullptr;            // ... its banks (made when the first batch is launched)
    rt_frame* ahead_owner = nullptr;   // this frame IS a bank of that frame
    uint32_t* frame_counts = nullptr; uint32_t* frame_slow = nullptr;              // k_frame's per-wave rows and slow-ray lists
    uint2* frame_spill = nullptr;                                                  // ... and its blocks' stack spill area
    uint32_t frame_blocks = 0, frame_chunks_per_wave = 0;
    uint64_t frame_launches = 0;                                                   // samples rendered by k_frame so far (rt_stats)
    // RT_OPT_FRAME_KERNEL = 255: the choice is MEASURED -- k_frame wins by 1.4 - 1.8 x on scenes of up to ~1 M triangles and loses 7 - 10 % on the 2.8 M /
    // 10 M ones (profiles/r05_call13.log), so the first frames of a scene time both: frames 0 - 1 (stage kernels) and 2 - 3 (k_frame) warm up, frames 4 - 19
    // ALTERNATE between the two (the device's clocks ramp over a process's first frames: timed in two blocks, whichever came second looked faster --
    // rt_render --frames chose k_frame for the 2.8 M-triangle scene, profiles/r05_call18.log) with HIP events around each frame's launches on the frame's
    // stream, read once the last has completed; then the faster way stays.
    struct { int frames = 0, timing = -1; bool decided = false, use_kernel = false, skip = false; uint64_t scene = 0; hipEvent_t ev[16][2] = {}; float ms_stage = 0.0f, ms_kernel = 0.0f; } fk_auto;
    uint32_t n_pipes = 1;          // pipes the current allocation holds
    uint32_t slots = 1;            // samples traced concurrently (resolved from slots_opt)
    uint32_t slots_opt = 0;        // RT_OPT_SAMPLES_IN_FLIGHT as set by the caller (0 = auto)
    uint32_t slots_limit = 0;      // != 0: a larger batch did not fit into device memory
    bool reserved_explicitly = false;  // rt_frame_reserve_samples has been called: rt_integrate does not second-guess the buffers' size
    uint64_t samples_asked = 0;        // samples rt_integrate has been asked for since the frame was made (lean growth of the buffers)
    uint32_t log_stride = 0;       // elements per log entry row = slots * chunk_pixels
    // Chunks also bound memory: RT_OPT_PATH_STATE_LIMIT_MB caps the per-path buffers of all pipes together
    // (path ids are chunk-relative, the radiance log is replayed per chunk).
    uint32_t chunk_pixels = 0;     // pixels per chunk as allocated (n_local when the tile is not chunked)
    uint32_t state_limit_mb = 0;   // RT_OPT_PATH_STATE_LIMIT_MB (0 = only the built-in 144 GB rule)
    uint32_t log_entries = 0;      // entries a path may log: 2 * (max_bounces + 1)
    // Radiance-log layout (DLog): compact = log_inline rows for every path + a pool of log_ovf_blocks overflow blocks;
    // full = every row for every path (log_inline == log_entries, no pool).
    uint32_t log_inline = 0, log_ovf_blocks = 0;
    uint32_t compact_log_opt = 2;  // RT_OPT_COMPACT_LOG: 0 = always the full layout, 1 = compact for batches of >= 8 samples in flight,
                                   // 2 (default) = compact when the caller bounds the path state (RT_OPT_PATH_STATE_LIMIT_MB)
    uint32_t log_pool_div = 8;     // the pool holds paths / log_pool_div blocks (RT_OPT_DEBUG_LOG_POOL_DIV: test hook)
    bool log_full_forced = false;  // a batch ran its pool dry: this frame stays on the full layout (until bounces / scene change)
    uint32_t fallback_limit_mb = 0;   // ... within the bytes the compact layout held (chunk_plan)
    uint32_t log_fallbacks = 0;    // batches repeated in the full layout
    bool fused = false;            // inside rt_integrate: whole samples, nothing reads the radiance between stages
    uint32_t trace_blocks;       // v1 grid
    uint32_t trace_variant = 5;  // RT_OPT_TRACE_VARIANT (5 = auto)
    uint64_t small_launch_paths = 3000000ull;   // RT_OPT_SMALL_LAUNCH_PATHS: launches of fewer rays run k_trace_w4 in chunk mode
    bool small_launch_set = false;              // ... set by the caller (otherwise the loop-D instance uses 8 M: launch_trace_w4)
    uint32_t trace_waves_per_cu = 0;   // RT_OPT_TRACE_WAVES_PER_CU (0 = LDS-limited residency)
    uint32_t chunk_refill = 1;                 // RT_OPT_CHUNK_REFILL: chunk mode refills idle lanes from the wave's own chunks
    uint64_t trace_tail_paths = 100000000ull; // RT_OPT_TRACE_TAIL_PATHS: batches of fewer paths launch the instance with loop D and refilled chunks (8 / 16 / 32 / 64 /
                                              // 128 samples of a 1080p frame in flight: +8 / +5 / +2.3 / -1.6 / -2.8 %, profiles/r04_call20_21.log, r04_call22.log)
    uint32_t trace_tail_lanes = 40;    // RT_OPT_TRACE_TAIL_LANES: k_trace_w4's loop D (0 = off); sweep: profiles/r04_call04_kernel_ab.log
    uint32_t select_form_box = 0;      // RT_OPT_TRACE_SELECT_FORM_BOX: every ray takes the select-form slab test
    uint32_t trace_tune = 0;           // RT_OPT_TRACE_TUNE: k_trace2 loop thresholds (0 = defaults)
    uint32_t timeline = 0;             // rt_frame_debug_timeline armed: k_trace_w4<closest> records its launch timeline
    uint32_t timeline_bounce = 0;
    uint32_t overlap_shadow = 1;       // RT_OPT_OVERLAP_SHADOW
    bool side_active = false;          // shadow traces go to PathPipe::side (set per stage call: side_on())
    // where the next trace launch goes (set by rt_intersect / rt_intersect_shadow)
    hipStream_t tl_stream = nullptr; uint2* tl_spill = nullptr; uint32_t* tl_slow_list = nullptr; uint32_t tl_flavour = 0;
    uint32_t shade_partition = 3;      // RT_OPT_SHADE_PARTITION: bit 0: k_shade sorts each block's entries hits first / misses last; bit 1: groups its output rays by octant
    uint32_t debug_alloc_limit = 0;    // RT_OPT_DEBUG_ALLOC_LIMIT: allocations above this many samples in flight fail
    // integrator state
    rt_camera camera;
    rt_camera camera_last;        // Integrator::prev_camera_ (integrator.hpp:89)
    rt_camera prev_camera;        // the kPrevCamera argument bound by the last rt_set_camera
    uint32_t sampler = 0;         // RT_OPT_SAMPLER: 0 kRandom, 1 kBlueNoise
    uint32_t aov = 0;             // RT_OPT_AOV
    uint32_t denoiser = 0;        // RT_OPT_DENOISER
    DAov aov_buf = {nullptr, nullptr, nullptr, nullptr};
    float4* prev_radiance = nullptr; float* prev_depth = nullptr;
    uint32_t max_bounces = 3;
    uint32_t white_furnace = 0;
    uint32_t drop_last = 1;
    uint32_t sample_count = 0;
    // RT_OPT_PROFILE_KERNELS
    uint32_t profile = 0;
    struct Span { hipEvent_t a, b; int cls; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> event_pool;
};

static int sync_frame_streams(rt_frame* f);
static void ahead_discard(rt_frame* f);      // RT_OPT_SAMPLES_AHEAD: whatever was traced ahead is dropped (reset, another camera, another option, a peek)
static void ahead_destroy(rt_frame* f);
static bool ahead_wanted(const rt_frame* f);
static int ahead_holds(const rt_frame* f, uint32_t sample);

namespace
{
int fail(rt_ctx* ctx, const std::string& msg)
{
    if (ctx) ctx->error = msg;
    g_thread_error = msg;
    return RT_ERROR;
}

#define HIPCHK(ctx, expr)                                                                         \
    do                                                                                            \
    {                                                                                             \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(ctx, std::string(#expr) + ": " + hipGetErrorString(e_));                  \
    } while (0)

int dev_alloc_copy(rt_ctx* ctx, void** out, const void* src, size_t bytes)
{
    *out = nullptr;
    size_t alloc = bytes ? bytes : 16;
    HIPCHK(ctx, hipMalloc(out, alloc));
    if (bytes && src) HIPCHK(ctx, hipMemcpyAsync(*out, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return RT_OK;
}

void free_scene(Scene& s)
{
    void* ptrs[] = {s.nodes, s.tris_rt, s.tris_sh, s.materials, s.textures, s.texture_data, s.lights, s.env, s.emissive, s.wnodes, s.mat_tex16, s.wnodes_sh, s.wnodes_cl};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (s.adapt) drop_fold_adapt(s.adapt);
    s = Scene();
}
} // namespace

extern "C" {

const char* rt_last_error(rt_ctx* ctx)
{
    return ctx ? ctx->error.c_str() : g_thread_error.c_str();
}

int rt_ctx_create(int device_ordinal, rt_ctx** out)
{
    if (!out) return fail(nullptr, "rt_ctx_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return fail(nullptr, std::string("rt_ctx_create: no HIP device (") + hipGetErrorString(e) + ")");
    if (device_ordinal < 0 || device_ordinal >= n) return fail(nullptr, "rt_ctx_create: bad device ordinal");
    rt_ctx* ctx = new rt_ctx;
    ctx->device = device_ordinal;
    if (hipSetDevice(device_ordinal) != hipSuccess || hipGetDeviceProperties(&ctx->prop, device_ordinal) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess)
    {
        delete ctx;
        return fail(nullptr, "rt_ctx_create: device initialisation failed");
    }
    if (hipMalloc((void**)&ctx->gamma_lut, 256 * sizeof(float)) != hipSuccess)
    {
        delete ctx;
        return fail(nullptr, "rt_ctx_create: out of device memory");
    }
    hipLaunchKernelGGL(k_fill_gamma_lut, dim3(1), dim3(256), 0, ctx->stream, ctx->gamma_lut);
    *out = ctx;
    return RT_OK;
}

int rt_ctx_destroy(rt_ctx* ctx)
{
    if (!ctx) return RT_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    free_scene(ctx->scene);
    if (ctx->blue_noise) (void)hipFree(ctx->blue_noise);
    if (ctx->gamma_lut) (void)hipFree(ctx->gamma_lut);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return RT_OK;
}

int rt_finish(rt_ctx* ctx)
{
    if (!ctx) return fail(nullptr, "rt_finish: ctx is NULL");
    (void)hipSetDevice(ctx->device);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    // ... and for whatever the frames put on streams of their own: chunks on other pipes, shadow traces on the side streams
    // (the stage API sends them there too since round 3) -- Finish() means everything (cl_context.cpp:115-118)
    // (not for the banks of RT_OPT_SAMPLES_AHEAD: what they trace ahead is the library's own business until a later Integrate() consumes it)
    for (rt_frame* f : ctx->frames)
        if (!f->ahead_owner && sync_frame_streams(f) != RT_OK) return RT_ERROR;
    return RT_OK;
}

int rt_ctx_device_info(rt_ctx* ctx, char* name, size_t name_len, int* compute_units, size_t* hbm_bytes)
{
    if (!ctx) return fail(nullptr, "rt_ctx_device_info: ctx is NULL");
    if (name && name_len) snprintf(name, name_len, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
    if (compute_units) *compute_units = ctx->prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = ctx->prop.totalGlobalMem;
    return RT_OK;
}

void* rt_ctx_stream(rt_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int rt_host_register(rt_ctx* ctx, void* host_ptr, size_t bytes)
{
    if (!ctx || !host_ptr || bytes == 0) return fail(ctx, "rt_host_register: bad argument");
    (void)hipSetDevice(ctx->device);
    HIPCHK(ctx, hipHostRegister(host_ptr, bytes, hipHostRegisterDefault));
    return RT_OK;
}

int rt_host_unregister(rt_ctx* ctx, void* host_ptr)
{
    if (!ctx || !host_ptr) return fail(ctx, "rt_host_unregister: bad argument");
    (void)hipSetDevice(ctx->device);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipHostUnregister(host_ptr));
    return RT_OK;
}

// The sampler tables CLPathTraceIntegrator uploads in its ctor (cl_pt_integrator.cpp:222-235)
int rt_upload_blue_noise_tables(rt_ctx* ctx, const int* sobol_256spp_256d, const int* scramblingTile, const int* rankingTile)
{
    if (!ctx || !sobol_256spp_256d || !scramblingTile || !rankingTile)
        return fail(ctx, "rt_upload_blue_noise_tables: NULL argument");
    (void)hipSetDevice(ctx->device);
    std::vector<uint8_t> packed(65536 + 131072 + 131072);
    const int* src[3] = {sobol_256spp_256d, scramblingTile, rankingTile};
    const size_t n[3] = {65536, 131072, 131072};
    size_t o = 0;
    for (int t = 0; t < 3; ++t)
        for (size_t i = 0; i < n[t]; ++i)
        {
            if (src[t][i] < 0 || src[t][i] > 255) return fail(ctx, "rt_upload_blue_noise_tables: table value outside 0..255");
            packed[o++] = (uint8_t)src[t][i];
        }
    if (!ctx->blue_noise) HIPCHK(ctx, hipMalloc((void**)&ctx->blue_noise, packed.size()));
    HIPCHK(ctx, hipMemcpy(ctx->blue_noise, packed.data(), packed.size(), hipMemcpyHostToDevice));
    return RT_OK;
}

int rt_ctx_set_option(rt_ctx* ctx, int option, uint32_t value)
{
    if (!ctx) return fail(nullptr, "rt_ctx_set_option: ctx is NULL");
    if (option == RT_CTX_OPT_TREELET_NODES)
    {
        if (value == 0 || value > 4096) return fail(ctx, "rt_ctx_set_option: treelet size must be 1..4096");
        ctx->treelet_nodes = value;
        return RT_OK;
    }
    if (option == RT_CTX_OPT_WIDE_BVH) { ctx->build_wide = value > 2u ? 1u : value; return RT_OK; }
    if (option == RT_CTX_OPT_SHADOW_TREE) { ctx->shadow_tree = value > 3u ? 1u : value; return RT_OK; }
    if (option == RT_CTX_OPT_CLOSEST_TREE) { ctx->closest_tree = value > 2u ? 1u : value; return RT_OK; }
    if (option == RT_CTX_OPT_ADAPTIVE_FOLD) { ctx->adaptive_fold = value & 31u; return RT_OK; }
    if (option == RT_CTX_OPT_DEVICE_FOLD) { ctx->device_fold = value ? 1u : 0u; return RT_OK; }
    if (option == RT_CTX_OPT_WIDE_LAYOUT) { ctx->wide_layout = value ? 1u : 0u; return RT_OK; }
    if (option == RT_CTX_OPT_TREE_BUILDER) { ctx->tree_builder = value > 2u ? 2u : value; return RT_OK; }
    if (option == RT_CTX_OPT_ADAPT_WAIT)
    {
        if (ctx->scene.adapt) fold_adapt_set_wait(ctx->scene.adapt, value != 0u);          // the scene in place only
        return RT_OK;
    }
    if (option == RT_CTX_OPT_ADAPT_MIN_INTERVAL_MS)
    {
        ctx->adapt_min_interval_ms = value;
        if (ctx->scene.adapt) fold_adapt_set_interval(ctx->scene.adapt, value);     // the scene in place too
        return RT_OK;
    }
    return fail(ctx, "rt_ctx_set_option: unknown option");
}

// ---- buffers ---------------------------------------------------------------
int rt_buffer_create(rt_ctx* ctx, size_t bytes, const void* init, rt_buffer** out)
{
    if (!ctx || !out) return fail(ctx, "rt_buffer_create: NULL argument");
    *out = nullptr;
    (void)hipSetDevice(ctx->device);
    void* p = nullptr;
    HIPCHK(ctx, hipMalloc(&p, bytes ? bytes : 16));
    if (init && bytes)
    {
        hipError_t e = hipMemcpy(p, init, bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(p); return fail(ctx, "rt_buffer_create: upload failed"); }
    }
    *out = new rt_buffer{ctx, p, bytes};
    return RT_OK;
}

int rt_buffer_destroy(rt_buffer* buf)
{
    if (!buf) return RT_OK;
    (void)hipStreamSynchronize(buf->ctx->stream);
    (void)hipFree(buf->ptr);
    delete buf;
    return RT_OK;
}

int rt_buffer_write(rt_buffer* buf, size_t offset, const void* src, size_t bytes)
{
    if (!buf || !src) return fail(nullptr, "rt_buffer_write: NULL argument");
    if (offset + bytes > buf->bytes) return fail(buf->ctx, "rt_buffer_write: out of range");
    HIPCHK(buf->ctx, hipMemcpyAsync((char*)buf->ptr + offset, src, bytes, hipMemcpyHostToDevice, buf->ctx->stream));
    HIPCHK(buf->ctx, hipStreamSynchronize(buf->ctx->stream));   // blocking, like WriteBuffer (CL_TRUE)
    return RT_OK;
}

int rt_buffer_read(rt_buffer* buf, size_t offset, void* dst, size_t bytes)
{
    if (!buf || !dst) return fail(nullptr, "rt_buffer_read: NULL argument");
    if (offset + bytes > buf->bytes) return fail(buf->ctx, "rt_buffer_read: out of range");
    HIPCHK(buf->ctx, hipMemcpyAsync(dst, (char*)buf->ptr + offset, bytes, hipMemcpyDeviceToHost, buf->ctx->stream));
    HIPCHK(buf->ctx, hipStreamSynchronize(buf->ctx->stream));
    return RT_OK;
}

int rt_buffer_copy(rt_buffer* src, rt_buffer* dst, size_t src_offset, size_t dst_offset, size_t bytes)
{
    if (!src || !dst) return fail(nullptr, "rt_buffer_copy: NULL argument");
    if (src_offset + bytes > src->bytes || dst_offset + bytes > dst->bytes)
        return fail(src->ctx, "rt_buffer_copy: out of range");
    HIPCHK(src->ctx, hipMemcpyAsync((char*)dst->ptr + dst_offset, (char*)src->ptr + src_offset, bytes,
        hipMemcpyDeviceToDevice, src->ctx->stream));
    return RT_OK;
}

void* rt_buffer_device_ptr(rt_buffer* buf) { return buf ? buf->ptr : nullptr; }
size_t rt_buffer_size(rt_buffer* buf) { return buf ? buf->bytes : 0; }

} // extern "C"

namespace
{
#include "fold_adapt_impl.h"

} // namespace

extern "C" {

// ---- scene -----------------------------------------------------------------
int rt_scene_upload(rt_ctx* ctx, const rt_scene_desc* sd)
{
    if (!ctx || !sd) return fail(ctx, "rt_scene_upload: NULL argument");
    if (!sd->triangles || sd->num_triangles == 0) return fail(ctx, "rt_scene_upload: no triangles");
    if (!sd->nodes || sd->num_nodes == 0) return fail(ctx, "rt_scene_upload: no BVH nodes");
    if (!sd->materials || sd->num_materials == 0) return fail(ctx, "rt_scene_upload: no materials");
    if (!sd->env_rgba || sd->env_width == 0 || sd->env_height == 0)
        return fail(ctx, "rt_scene_upload: no environment image");
    (void)hipSetDevice(ctx->device);
    // nothing may still read the scene that is about to be freed: batches traced ahead are dropped (RT_OPT_SAMPLES_AHEAD), and every
    // stream a frame launches on -- side streams, pipes, banks -- has drained
    for (rt_frame* f : ctx->frames) ahead_discard(f);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (rt_frame* f : ctx->frames)
        if (sync_frame_streams(f) != RT_OK) return RT_ERROR;
    free_scene(ctx->scene);
    ++ctx->scene_uploads;
    Scene& s = ctx->scene;
    const uint32_t nt = sd->num_triangles, nn = sd->num_nodes;
    const auto t_upload = std::chrono::steady_clock::now();
    auto lap = [](std::chrono::steady_clock::time_point& t) { const auto n = std::chrono::steady_clock::now(); const double d = std::chrono::duration<double>(n - t).count(); t = n; return d; };
    auto t_lap = t_upload;
    double t_layout = 0.0, t_device = 0.0, t_fold = 0.0, t_own_wait = 0.0, t_choose = 0.0;
    // the trees of the backend's own (below, "Trees of the backend's own") are built on worker threads meanwhile
    OwnTree own_sh, own_cl, own_sh_dev;      // own_sh_dev: the shadow rays' candidate built on the device (RT_CTX_OPT_TREE_BUILDER != 0)
    const bool may_own = ctx->build_wide == 1u && (uint64_t)nt * 64 <= 0xFFFFFFFFull && (sd->nodes[0].num_primitives_axis >> 16) == 0;
    if (ctx->device_fold && ctx->build_wide == 1u) { own_sh.device = ctx->device; own_cl.device = ctx->device; }
    own_sh.pairs = own_cl.pairs = ctx->wide_layout != 0u;
    // (the shadow rays' tree only: the closest-hit rays' own tree is the tolerance mode's, host-built)
    const bool try_device_tree = may_own && ctx->shadow_tree && ctx->tree_builder != 0u && ctx->device_fold != 0u && ctx->build_wide == 1u;
    if (try_device_tree)
    {
        own_sh_dev.device = ctx->device; own_sh_dev.pairs = own_sh.pairs; own_sh_dev.device_builder = true; own_sh_dev.device_only = true;
        own_sh_dev.start(sd, true, ctx->shadow_tree);
    }
    if (may_own && ctx->shadow_tree && !(try_device_tree && ctx->tree_builder == 1u)) own_sh.start(sd, true, ctx->shadow_tree);
    if (may_own && ctx->closest_tree) own_cl.start(sd, false, ctx->closest_tree);
    // what an adaptation keeps of the caller's arrays (FoldAdapt, below: the binary tree, the triangles' corners -- 157 + 100 MB for 2.8 M triangles) is copied beside
    // everything else instead of after it
    struct AdaptCopies
    {
        std::vector<rt_bvh_node> bvh2; std::vector<float> tri9; std::thread worker;
        ~AdaptCopies() { if (worker.joinable()) worker.join(); }
    } adapt_copies;
    if ((ctx->adaptive_fold & 1u) && ctx->build_wide == 1u && !ctx->closest_tree && (nn >= 8192u || (ctx->adaptive_fold & 4u)))
        adapt_copies.worker = std::thread([&adapt_copies, sd, nn, nt, corners = (ctx->adaptive_fold & 16u) != 0u]()
        {
            adapt_copies.bvh2.assign(sd->nodes, sd->nodes + nn);
            if (!corners) return;
            adapt_copies.tri9.resize((size_t)nt * 9);
            for (uint32_t i = 0; i < nt; ++i)
            {
                const rt_triangle& t = sd->triangles[i];
                const rt_float3 v[3] = {t.v1.position, t.v2.position, t.v3.position};
                for (int k = 0; k < 3; ++k) { adapt_copies.tri9[(size_t)i * 9 + 3 * k] = v[k].x; adapt_copies.tri9[(size_t)i * 9 + 3 * k + 1] = v[k].y; adapt_copies.tri9[(size_t)i * 9 + 3 * k + 2] = v[k].z; }
            }
        });

    // --- BVH re-layout: LinearBVHNode[] (bvh.cpp:223-245) -> child-pair records, in treelet order (treelet_order.h: a pure permutation of records)
    std::vector<uint32_t> interior_index;
    uint32_t n_interior = 0;
    {
        const int bad = treelet::order(sd->nodes, nn, ctx->treelet_nodes, interior_index, n_interior);
        if (bad == 1) return fail(ctx, "rt_scene_upload: child index outside the node array");
        if (bad != 0) return fail(ctx, "rt_scene_upload: the node array is not a tree (cycle)");
    }
    t_layout = lap(t_lap);
    // the records themselves are written on the device (k_relayout_*), below
    std::vector<float4> super_root(4);
    const rt_bvh_node& root = sd->nodes[0];
    s.d.root_ref = (root.num_primitives_axis >> 16) > 0 ? (RT_LEAF_BIT | root.offset) : 0u;
    {
        // super-root: child 0 = (root box, root ref), child 1 = empty.  Visiting it IS the
        // reference's first loop iteration (box test of node 0, trace_bvh.cl:146-148).
        s.d.entry_ref = n_interior;
        float4* out = super_root.data();
        out[0] = make_float4(root.bounds_min.x, root.bounds_min.y, root.bounds_max.x, root.bounds_max.y);
        out[1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        out[2] = make_float4(root.bounds_min.z, root.bounds_max.z, 0.0f, 0.0f);
        uint32_t r0 = s.d.root_ref, r1 = RT_EMPTY_REF, axis = 0;
        float fr0, fr1, fax;
        memcpy(&fr0, &r0, 4); memcpy(&fr1, &r1, 4); memcpy(&fax, &axis, 4);
        out[3] = make_float4(fr0, fr1, fax, 0.0f);
    }
    s.d.root_min[0] = root.bounds_min.x; s.d.root_min[1] = root.bounds_min.y; s.d.root_min[2] = root.bounds_min.z;
    s.d.root_max[0] = root.bounds_max.x; s.d.root_max[1] = root.bounds_max.y; s.d.root_max[2] = root.bounds_max.z;

    std::vector<float4> lights((size_t)(sd->num_lights ? sd->num_lights : 1) * 3);
    for (uint32_t i = 0; i < sd->num_lights; ++i)
    {
        const rt_light& l = sd->lights[i];
        float ft; uint32_t ty = l.type; memcpy(&ft, &ty, 4);
        lights[(size_t)i * 3 + 0] = make_float4(l.origin.x, l.origin.y, l.origin.z, 0.0f);
        lights[(size_t)i * 3 + 1] = make_float4(l.radiance.x, l.radiance.y, l.radiance.z, 0.0f);
        lights[(size_t)i * 3 + 2] = make_float4(ft, 0.0f, 0.0f, 0.0f);
    }
    {
        // shadow rays towards a directional light all share its direction (light.h:57-61: origin * 20000, normalised again
        // by HitSurface): a zero or tiny component makes 1/dir non-finite (RT_SIGN_SLOW) for every one of them
        uint32_t slow_lights = 0;
        for (uint32_t i = 0; i < sd->num_lights; ++i)
        {
            const rt_light& l = sd->lights[i];
            if (l.type == RT_LIGHT_TYPE_POINT) continue;
            const double len = std::sqrt((double)l.origin.x * l.origin.x + (double)l.origin.y * l.origin.y + (double)l.origin.z * l.origin.z);
            const double lim = len * 0x1p-95;
            if (!(std::fabs((double)l.origin.x) > lim && std::fabs((double)l.origin.y) > lim && std::fabs((double)l.origin.z) > lim)) ++slow_lights;
        }
        s.slow_shadow = slow_lights != 0 && 4u * slow_lights >= sd->num_lights;
    }
    for (uint32_t i = 0; i < sd->num_materials; ++i)
    {
        // every 8-bit texture slot of a packed material: 0xFF = none (constants.h:35), else an index into textures
        const rt_packed_material& m = sd->materials[i];
        const uint32_t idx[6] = {m.diffuse_albedo >> 24, m.specular_albedo >> 24, (m.roughness_metalness >> 8) & 0xFFu,
            m.roughness_metalness >> 24, (m.ior_emission_idx_transparency >> 8) & 0xFFu, m.ior_emission_idx_transparency >> 24};
        if (!sd->material_texture_indices)
            for (uint32_t t : idx)
                if (t != RT_INVALID_TEXTURE_IDX && t >= sd->num_textures)
                    return fail(ctx, "rt_scene_upload: material references a texture that does not exist");
    }
    if (sd->material_texture_indices)          // the wide indices replace the packed ones (rt_scene_desc)
        for (size_t i = 0; i < (size_t)sd->num_materials * 6; ++i)
            if (sd->material_texture_indices[i] != 0xFFFFu && sd->material_texture_indices[i] >= sd->num_textures)
                return fail(ctx, "rt_scene_upload: material_texture_indices references a texture that does not exist");
    for (uint32_t i = 0; i < sd->num_emissive; ++i)
        if (sd->emissive_indices[i] >= nt) return fail(ctx, "rt_scene_upload: emissive index outside the triangle array");
    for (uint32_t i = 0; i < sd->num_textures; ++i)
    {
        const rt_texture& t = sd->textures[i];
        if (t.width <= 0 || t.height <= 0 || t.data_start < 0 ||
            (uint64_t)t.data_start + (uint64_t)t.width * t.height > sd->num_texture_data)
            return fail(ctx, "rt_scene_upload: texture outside texture_data");
    }

    // --- re-layout on the device: the reference's arrays go to HBM as they are, three streaming
    // kernels write the child-pair node records and the 64-byte / 128-byte triangle records
    int rc = RT_OK;
    void *raw_tris = nullptr, *raw_nodes = nullptr, *d_index = nullptr, *d_last = nullptr, *d_err = nullptr;
    rc |= dev_alloc_copy(ctx, &raw_tris, sd->triangles, (size_t)nt * sizeof(rt_triangle));
    rc |= dev_alloc_copy(ctx, &raw_nodes, sd->nodes, (size_t)nn * sizeof(rt_bvh_node));
    rc |= dev_alloc_copy(ctx, &d_index, interior_index.data(), (size_t)nn * sizeof(uint32_t));
    rc |= dev_alloc_copy(ctx, &d_last, nullptr, (size_t)nt);
    rc |= dev_alloc_copy(ctx, &d_err, nullptr, sizeof(int));
    rc |= dev_alloc_copy(ctx, &s.nodes, nullptr, (size_t)(n_interior + 1) * 64);   // + the super-root record
    rc |= dev_alloc_copy(ctx, &s.tris_rt, nullptr, (size_t)nt * 64);
    rc |= dev_alloc_copy(ctx, &s.tris_sh, nullptr, (size_t)nt * 128);
    auto free_temps = [&]() { for (void* p : {raw_tris, raw_nodes, d_index, d_last, d_err}) if (p) (void)hipFree(p); };
    if (rc != RT_OK) { free_temps(); free_scene(s); return RT_ERROR; }
    int relayout_err = RL_OK;
    bool ok = hipMemsetAsync(d_last, 0, (size_t)nt, ctx->stream) == hipSuccess &&
              hipMemsetAsync(d_err, 0, sizeof(int), ctx->stream) == hipSuccess;
    if (ok)
    {
        hipLaunchKernelGGL(k_relayout_mark_leaves, dim3((nn + 255u) / 256u), dim3(256), 0, ctx->stream,
            (const rt_bvh_node*)raw_nodes, nn, nt, (uint8_t*)d_last, (int*)d_err);
        hipLaunchKernelGGL(k_relayout_nodes, dim3((nn + 255u) / 256u), dim3(256), 0, ctx->stream,
            (const rt_bvh_node*)raw_nodes, nn, (const uint32_t*)d_index, (float4*)s.nodes, (int*)d_err);
        hipLaunchKernelGGL(k_relayout_triangles, dim3((nt + 255u) / 256u), dim3(256), 0, ctx->stream,
            (const rt_triangle*)raw_tris, nt, sd->num_materials, (const uint8_t*)d_last, (float4*)s.tris_rt,
            (float4*)s.tris_sh, (int*)d_err);
        hipLaunchKernelGGL(k_relayout_leaf_bounds, dim3((nn + 255u) / 256u), dim3(256), 0, ctx->stream,
            (const rt_bvh_node*)raw_nodes, nn, nt, (float4*)s.tris_rt);
        ok = hipGetLastError() == hipSuccess &&
             hipMemcpyAsync((char*)s.nodes + (size_t)n_interior * 64, super_root.data(), 64, hipMemcpyHostToDevice,
                 ctx->stream) == hipSuccess &&
             hipMemcpyAsync(&relayout_err, d_err, sizeof(int), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
             hipStreamSynchronize(ctx->stream) == hipSuccess;
    }
    // RT_CTX_OPT_DEVICE_FOLD: the reference tree's collapse into 4-wide records, from the node array the re-layout kernels have just read (device_fold.h)
    std::vector<WideNode> wide;
    uint32_t w_entry = 0;
    std::vector<uint32_t> wide_roots;
    WideNode* d_wide = nullptr;
    uint32_t n_wide_dev = 0;
    bool folded_on_device = false;
    double t_dev_fold = 0.0;
    if (ok && relayout_err == RL_OK && ctx->device_fold && ctx->build_wide == 1u && (uint64_t)nt * 64 <= 0xFFFFFFFFull)
    {
        const bool want_host_copy = ctx->wide_layout != 0u || (may_own && (ctx->shadow_tree == 1u || ctx->closest_tree == 1u));   // the pair layout and the choice by proxy rays work on the host
        folded_on_device = devfold::fold(ctx->stream, (const rt_bvh_node*)raw_nodes, nn, sd->nodes[0], nullptr, nullptr, &d_wide, &n_wide_dev, &w_entry, &wide_roots,
                                         want_host_copy ? &wide : nullptr, nullptr, &t_dev_fold);
        (void)hipGetLastError();
    }
    free_temps();
    if (!ok) { if (d_wide) (void)hipFree(d_wide); free_scene(s); return fail(ctx, "rt_scene_upload: device re-layout failed"); }
    if (relayout_err != RL_OK)
    {
        if (d_wide) (void)hipFree(d_wide);
        free_scene(s);
        switch (relayout_err)
        {
        case RL_CHILD_RANGE: return fail(ctx, "rt_scene_upload: child index outside the node array");
        case RL_LEAF_RANGE: return fail(ctx, "rt_scene_upload: leaf range outside the triangle array");
        case RL_AXIS: return fail(ctx, "rt_scene_upload: bad split axis");
        default: return fail(ctx, "rt_scene_upload: material index out of range");
        }
    }
    rc |= dev_alloc_copy(ctx, &s.materials, sd->materials, (size_t)sd->num_materials * sizeof(rt_packed_material));
    rc |= dev_alloc_copy(ctx, &s.textures, sd->textures, (size_t)sd->num_textures * sizeof(rt_texture));
    rc |= dev_alloc_copy(ctx, &s.texture_data, sd->texture_data, (size_t)sd->num_texture_data * 4);
    rc |= dev_alloc_copy(ctx, &s.lights, lights.data(), lights.size() * sizeof(float4));
    rc |= dev_alloc_copy(ctx, &s.env, sd->env_rgba, (size_t)sd->env_width * sd->env_height * 16);
    rc |= dev_alloc_copy(ctx, &s.emissive, sd->emissive_indices, (size_t)sd->num_emissive * 4);
    if (sd->material_texture_indices)
        rc |= dev_alloc_copy(ctx, &s.mat_tex16, sd->material_texture_indices, (size_t)sd->num_materials * 6 * sizeof(uint16_t));
    t_device = lap(t_lap) - t_dev_fold;
    // the 4-wide quantized tree for k_trace_w4 (optional: trees that do not qualify keep the BVH2 kernels)
    // (a SAH collapse can be deeper than the kernel's stack bound allows where two levels at a time are not: try both)
    const bool have_wide = folded_on_device || (ctx->build_wide && (uint64_t)nt * 64 <= 0xFFFFFFFFull &&
        ((ctx->build_wide != 2u && build_wide_bvh(sd->nodes, nn, RT_WIDE_SAH, wide, w_entry, &wide_roots)) ||
         build_wide_bvh(sd->nodes, nn, RT_WIDE_TWO_LEVELS, wide, w_entry, &wide_roots)));
    const uint32_t n_wide_ref = folded_on_device ? n_wide_dev : (uint32_t)wide.size();
    if (have_wide && ctx->wide_layout && !wide.empty())
    {
        pair_layout_by_area(wide, wide_roots, sd->nodes, nn, (const ownbvh::Metric*)nullptr);
        if (folded_on_device && hipMemcpyAsync(d_wide, wide.data(), wide.size() * sizeof(WideNode), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc |= fail(ctx, "rt_scene_upload: uploading the paired records failed");
    }
    if (folded_on_device) s.wnodes = d_wide;
    else if (have_wide) rc |= dev_alloc_copy(ctx, &s.wnodes, wide.data(), wide.size() * sizeof(WideNode));
    // Trees of the backend's own over the reference's leaves (own_bvh.h): one for the shadow rays (exact: an any-hit verdict
    // does not depend on what sits above the leaves) and -- opt-in, tolerance mode -- one for the closest-hit rays.  Which
    // candidate a population walks is MEASURED with proxy rays (tree_select.h); the reference's own topology is a candidate,
    // so the choice is never worse than sharing on that measure.
    bool have_sh = false, have_cl = false;
    s.tree_report.clear();
    t_fold = lap(t_lap) + t_dev_fold;
    // the shadow rays' candidates: the one built on the device is ready first and is measured first; if it wins, the host's build is abandoned
    OwnTree* sh = &own_sh;
    ChoiceInputs shadow_choice;              // (the proxy rays and the reference fold's cost: measured once for both candidates)
    if (try_device_tree)
    {
        own_sh_dev.join();
        (void)hipSetDevice(ctx->device);
        if (own_sh_dev.ok && have_wide && n_wide_ref != 0u && choose_tree(sd, wide, w_entry, true, ctx->shadow_tree, own_sh_dev, s.tree_report, &shadow_choice))
        {
            sh = &own_sh_dev;
            own_sh.cancel_build.store(true);
        }
    }
    own_sh.join(); own_cl.join();
    t_own_wait = lap(t_lap);
    (void)hipSetDevice(ctx->device);
    auto adopt_own = [&](OwnTree& own, void*& dst)
    {
        if (own.d_wide) { dst = own.d_wide; own.d_wide = nullptr; }                        // folded on the device: the records are there
        else rc |= dev_alloc_copy(ctx, &dst, own.wide.data(), own.wide.size() * sizeof(WideNode));
    };
    if (have_wide && n_wide_ref != 0u && may_own && ctx->shadow_tree)
    {
        have_sh = sh == &own_sh_dev || choose_tree(sd, wide, w_entry, true, ctx->shadow_tree, own_sh, s.tree_report, &shadow_choice);
        if (have_sh) adopt_own(*sh, s.wnodes_sh);
    }
    if (have_wide && n_wide_ref != 0u && may_own && ctx->closest_tree)
    {
        have_cl = choose_tree(sd, wide, w_entry, false, ctx->closest_tree, own_cl, s.tree_report);
        if (have_cl) adopt_own(own_cl, s.wnodes_cl);
    }
    const uint32_t w_entry_sh = sh->entry, w_entry_cl = own_cl.entry;
    t_choose = lap(t_lap);
    if (rc != RT_OK) { free_scene(s); return RT_ERROR; }
    // RT_CTX_OPT_ADAPTIVE_FOLD: what the first rt_integrate needs to fold these trees again for its own rays (FoldAdapt)
    if ((ctx->adaptive_fold & 1u) && have_wide && n_wide_ref != 0u && !have_cl && (nn >= 8192u || (ctx->adaptive_fold & 4u)))
    {
        FoldAdapt* a = new FoldAdapt();
        a->mode = ctx->adaptive_fold;
        a->device = ctx->device;
        a->device_fold = ctx->device_fold != 0u && ctx->build_wide == 1u;
        a->pairs = ctx->wide_layout != 0u;
        a->min_interval_ms = ctx->adapt_min_interval_ms;
        memset(&a->camera, 0, sizeof(a->camera));
        {
            const double ex = (double)root.bounds_max.x - root.bounds_min.x, ey = (double)root.bounds_max.y - root.bounds_min.y, ez = (double)root.bounds_max.z - root.bounds_min.z;
            a->scene_diagonal = std::sqrt(ex * ex + ey * ey + ez * ez);
        }
        if (adapt_copies.worker.joinable()) adapt_copies.worker.join();
        if (adapt_copies.bvh2.size() == nn) a->bvh2.swap(adapt_copies.bvh2); else a->bvh2.assign(sd->nodes, sd->nodes + nn);
        a->roots = std::move(wide_roots);
        if (have_sh) { a->bvh2_sh = std::move(sh->bvh2); a->roots_sh = std::move(sh->roots); }
        if (a->mode.load() & 16u)
        {
            if (adapt_copies.tri9.size() == (size_t)nt * 9) a->tri9.swap(adapt_copies.tri9);
            else
            {
                a->tri9.resize((size_t)nt * 9);
                for (uint32_t i = 0; i < nt; ++i)
                {
                    const rt_triangle& t = sd->triangles[i];
                    const rt_float3 v[3] = {t.v1.position, t.v2.position, t.v3.position};
                    for (int k = 0; k < 3; ++k) { a->tri9[(size_t)i * 9 + 3 * k] = v[k].x; a->tri9[(size_t)i * 9 + 3 * k + 1] = v[k].y; a->tri9[(size_t)i * 9 + 3 * k + 2] = v[k].z; }
                }
            }
        }
        s.adapt = a;
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // host staging vectors die here

    s.d.nodes = (const float4*)s.nodes;
    s.d.tris_rt = (const float4*)s.tris_rt;
    s.d.tris_sh = (const float4*)s.tris_sh;
    s.d.materials = (const rt_packed_material*)s.materials;
    s.d.textures = (const rt_texture*)s.textures;
    s.d.texture_data = (const uint32_t*)s.texture_data;
    s.d.lights = (const float4*)s.lights;
    s.d.env = (const float4*)s.env;
    s.d.gamma_lut = ctx->gamma_lut;
    s.d.env_w = (int)sd->env_width;
    s.d.env_h = (int)sd->env_height;
    s.d.mat_tex16 = (const uint16_t*)s.mat_tex16;
    s.d.emissive = (const uint32_t*)s.emissive;
    s.d.emissive_count = sd->num_emissive;
    s.d.emissive_nee = (sd->flags & RT_SCENE_EMISSIVE_NEE) && sd->num_emissive ? 1u : 0u;
    s.d.light_count = sd->num_lights;
    s.d.wnodes = have_wide ? (const float4*)s.wnodes : nullptr;
    s.d.w_entry_ref = w_entry;
    s.d.wnodes_sh = have_sh ? (const float4*)s.wnodes_sh : s.d.wnodes;
    s.d.w_sh_entry_ref = have_sh ? w_entry_sh : w_entry;
    if (have_cl) { s.d.wnodes = (const float4*)s.wnodes_cl; s.d.w_entry_ref = w_entry_cl; }
    s.n_wide = have_wide ? n_wide_ref : 0u;
    s.n_tris = nt;
    s.n_wide_sh = have_sh ? (uint32_t)sh->wide.size() : 0u;
    s.n_wide_cl = have_cl ? (uint32_t)own_cl.wide.size() : 0u;
    s.wide_ok = have_wide;
    s.offsets32 = (uint64_t)(n_interior + 1) * 64 <= 0xFFFFFFFFull && (uint64_t)nt * 64 <= 0xFFFFFFFFull;
    if (!s.offsets32)
        fprintf(stderr, "rt_scene_upload: warning: the node or trace-triangle records reach 4 GiB (%u interior nodes, %u triangles): k_trace_w4 and k_trace2 "
                        "address them with 32-bit byte offsets, so every launch takes the per-ray kernel k_trace_v1 -- correct, and several times slower\n",
            n_interior, nt);
    s.valid = true;
    {
        // where the upload's time went (rt_scene_tree_report; bench.py prints it as `setup`)
        char line[400];
        const double t_rest = lap(t_lap);
        snprintf(line, sizeof(line), "upload: %.3f s = record order on the host %.3f + copies and re-layout kernels %.3f + fold of the reference's tree %.3f (%s) + waiting for the own tree(s) %.3f "
            "(shadow tree: built %s in %.3f, folded in %.3f) + choosing by proxy rays and uploading %.3f + adaptation state %.3f (%u triangles, %u nodes, %u + %u wide records)\n",
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t_upload).count(), t_layout, t_device, t_fold, folded_on_device ? "on the device" : "on host threads", t_own_wait,
            sh->built_on_device ? "on the device (PLOC)" : (try_device_tree ? "on host threads (the device-built candidate did not win its measurement)" : "on host threads"),
            sh->build_seconds, sh->fold_seconds, t_choose, t_rest, nt, nn, s.n_wide, s.n_wide_sh);
        s.tree_report += line;
    }
    return RT_OK;
}

// ---- frame -----------------------------------------------------------------
} // extern "C"

namespace
{
void free_path_buffers(rt_frame* f)
{
    // a shadow trace may still be running on a pipe's side stream (rt_integrate overlaps it with the next bounce, and an
    // rt_integrate that failed mid-bounce never waited for it): nothing is freed under it
    for (PathPipe& q : f->ps)
    {
        if (q.side) (void)hipStreamSynchronize(q.side);
        q.shadow_in_flight[0] = q.shadow_in_flight[1] = false;
    }
    for (PathPipe& q : f->ps)
    {
        void* ptrs[] = {q.o4[0], q.o4[1], q.d4[0], q.d4[1], q.thr[0], q.thr[1], q.hits, q.sh_o4[0], q.sh_d4[0],
            q.sh_aux[0], q.sh_o4[1], q.sh_d4[1], q.sh_aux[1], q.rlog, q.cnt, q.slow_list, q.sh_slow_list, q.ovf_slot};
        for (void* p : ptrs) if (p) (void)hipFree(p);
        for (int i = 0; i < 2; ++i)
        {
            q.o4[i] = nullptr; q.d4[i] = nullptr; q.thr[i] = nullptr;
            q.sh_o4[i] = nullptr; q.sh_d4[i] = nullptr; q.sh_aux[i] = nullptr;
        }
        q.hits = nullptr; q.rlog = nullptr; q.cnt = nullptr; q.slow_list = nullptr; q.sh_slow_list = nullptr; q.ovf_slot = nullptr;
    }
}

// Per-path state: ray queues for `slots` samples in flight and the radiance log
// with 2 * (max_bounces + 1) entries per path.  (Re)allocated when either changes.
// Compact radiance log (kernels_common.h, DLog): six inline entries per path + overflow blocks for an eighth of the paths.
#define RT_LOG_INLINE 6u
// may a batch of `slots` samples in flight use the compact layout?  (It pays from ~10 entries per path up, needs batches large
// enough for the pool's statistics, and one pipe: the pool-dry check synchronises the host with the chunk's stream.)
bool log_is_compact(const rt_frame* f, uint32_t slots)
{
    const bool asked = f->compact_log_opt == 1u || (f->compact_log_opt == 2u && f->state_limit_mb != 0u);
    return asked && !f->log_full_forced && slots >= 8u && f->pipelines == 1u && 2u * (f->max_bounces + 1u) >= RT_LOG_INLINE + 4u;
}

// ray queues o4, d4, thr (x2), hits, shadow queue o4, d4 (x2) = 11 x 16; sh_aux (x2), cnt, two slow lists = 5 x 4;
// log, full layout: 2 (B + 1) entries of 12 bytes -- 412 bytes at 8 bounces (rounds 1-2: 540); compact layout: 6 inline
// entries + the path's share of the overflow pool + its block index -- 290 bytes at 8 bounces, 314 at 16.
size_t bytes_per_path(const rt_frame* f, uint32_t slots)
{
    const size_t entries = 2u * (f->max_bounces + 1u);
    if (!log_is_compact(f, slots)) return 11u * 16u + 5u * 4u + 12u * entries;
    return 11u * 16u + 5u * 4u + 4u + 12u * RT_LOG_INLINE + (12u * (entries - RT_LOG_INLINE) + f->log_pool_div - 1u) / f->log_pool_div;
}

// auto: the most samples (<= 1024; a multiple of 8 above 8, of 16 above 64 -- rounds 1-2 took powers of two, which left up to
// half of the budget unused: a 4K frame with 16 bounces got 16 samples in flight where 24 fit) that keep tile pixels x
// samples inside 32-bit path ids and the per-path buffers under 144 GiB (half of the 288 GB of HBM)
uint32_t auto_slots(const rt_frame* f)
{
    const uint64_t n = f->n_local ? f->n_local : 1;
    const uint64_t max_paths = 0xFFFFFFF0ull;
    uint64_t by_memory = (144ull << 30) / (n * bytes_per_path(f, 1024u));        // the compact layout, if this frame may use it ...
    if (by_memory < 8u) by_memory = (144ull << 30) / (n * bytes_per_path(f, 1u));  // ... which takes 8 samples in flight
    const uint64_t by_ids = max_paths / n;
    uint64_t s = by_memory < by_ids ? by_memory : by_ids;
    // ... and no more than fill a launch: beyond ~270 M paths in flight (128 samples of a 1080p frame) a larger batch buys
    // nothing (176 in flight, 140 GB: 6112 Mrays/s; 128, 102 GB: 6195 -- profiles/r03_call02_bench.json, r03_mid_bench.json)
    const uint64_t by_fill = (270000000ull + n - 1) / n;
    if (s > by_fill) s = by_fill;
    if (s > 1024) s = 1024;
    if (s > 64) s &= ~15ull;
    else if (s > 8) s &= ~7ull;
    return s ? (uint32_t)s : 1u;
}

// the most samples rt_integrate will trace together
uint32_t slot_cap(const rt_frame* f)
{
    uint32_t cap = f->slots_opt ? f->slots_opt : auto_slots(f);
    return f->slots_limit && f->slots_limit < cap ? f->slots_limit : cap;    // what the device could actually hold
}

// How the tile is cut for `slots` samples in flight: number of pipes and pixels per chunk.
//  * two (RT_OPT_PIPELINES) pipes when a batch is large enough for its launches to fill the machine twice over
//    (>= 2 samples in flight and >= 4 M paths): the tile's halves then overlap each other's launch tails;
//  * RT_OPT_PATH_STATE_LIMIT_MB bounds the per-path buffers of all pipes together: more, smaller chunks
//    (multiples of 4096 pixels), still dealt round-robin to the pipes.
void chunk_plan(const rt_frame* f, uint32_t slots, uint32_t& n_pipes, uint32_t& chunk_pixels)
{
    const uint64_t n = f->n_local ? f->n_local : 1;
    slots = slots ? slots : 1u;
    n_pipes = 1;
    if (f->pipelines > 1 && slots >= 2 && n * slots >= 4000000ull) n_pipes = f->pipelines < RT_MAX_PIPES ? f->pipelines : RT_MAX_PIPES;
    // one sample per pixel in flight (RT_OPT_STAGE_PIPES): the interactive features need the whole tile in one chunk
    if (f->stage_pipes > 1 && slots == 1 && n >= 262144ull && !(f->denoiser || f->aov != 0)) n_pipes = f->stage_pipes < RT_MAX_PIPES ? f->stage_pipes : RT_MAX_PIPES;
    uint64_t c = (n + n_pipes - 1) / n_pipes;                        // pixels per chunk without a memory limit
    if (n_pipes > 1) c = (c + 63ull) & ~63ull;
    // the caller's limit, and -- after a batch had to fall back from the compact to the full log layout -- the library's own
    // 144 GiB rule (the full layout then runs the same batch size in more chunks instead of asking for more memory than that)
    const uint64_t limit_mb = f->state_limit_mb && f->fallback_limit_mb ? (f->state_limit_mb < f->fallback_limit_mb ? f->state_limit_mb : f->fallback_limit_mb)
                                                                         : (f->state_limit_mb ? f->state_limit_mb : f->fallback_limit_mb);
    if (limit_mb)
    {
        const uint64_t per_pixel = (uint64_t)slots * bytes_per_path(f, slots);
        const uint64_t limit = (limit_mb << 20) / n_pipes;
        if (c * per_pixel > limit)
        {
            c = (limit / per_pixel) & ~4095ull;
            if (c < 4096) c = 4096;
        }
    }
    chunk_pixels = (uint32_t)(c < n ? c : n);
    if ((uint64_t)chunk_pixels * n_pipes > n + chunk_pixels) n_pipes = (uint32_t)((n + chunk_pixels - 1) / chunk_pixels);
}

uint32_t chunk_for(const rt_frame* f, uint32_t slots)
{
    uint32_t np, c;
    chunk_plan(f, slots, np, c);
    return c;
}

int alloc_path_buffers(rt_frame* f, uint32_t slots)
{
    rt_ctx* ctx = f->ctx;
    free_path_buffers(f);
    // test hook (tests/test_gpu_headline_parity.py): pretend the device cannot hold more than N samples in flight
    if (f->debug_alloc_limit && slots > f->debug_alloc_limit)
        return fail(ctx, "out of device memory for the per-path buffers (RT_OPT_DEBUG_ALLOC_LIMIT)");
    f->slots = slots ? slots : 1u;
    chunk_plan(f, f->slots, f->n_pipes, f->chunk_pixels);
    f->p = &f->ps[0];
    uint64_t paths = (uint64_t)f->chunk_pixels * f->slots;
    if (paths > 0xFFFFFFF0ull) return fail(ctx, "samples in flight x tile pixels exceeds the 32-bit path-id range");
    f->log_stride = (uint32_t)paths;
    f->log_entries = 2u * (f->max_bounces + 1u);
    const bool compact = log_is_compact(f, f->slots);
    f->log_inline = compact ? RT_LOG_INLINE : f->log_entries;
    f->log_ovf_blocks = compact ? (uint32_t)(paths / f->log_pool_div > 64u ? paths / f->log_pool_div : 64u) : 0u;
    const size_t log_elems = (size_t)f->log_inline * paths + (size_t)(f->log_entries - f->log_inline) * f->log_ovf_blocks;
    size_t q = (size_t)(paths + 4) * sizeof(float4);
    bool ok = true;
    for (uint32_t i = 0; i < f->n_pipes && ok; ++i)
    {
        PathPipe& pp = f->ps[i];
        void** ptrs[] = {(void**)&pp.o4[0], (void**)&pp.o4[1], (void**)&pp.d4[0], (void**)&pp.d4[1],
            (void**)&pp.thr[0], (void**)&pp.thr[1], (void**)&pp.hits, (void**)&pp.sh_o4[0],
            (void**)&pp.sh_d4[0], (void**)&pp.sh_o4[1], (void**)&pp.sh_d4[1]};
        for (void** p : ptrs) ok = ok && hipMalloc(p, q) == hipSuccess;
        for (uint32_t*& p : pp.sh_aux) ok = ok && hipMalloc((void**)&p, (size_t)(paths + 4) * sizeof(uint32_t)) == hipSuccess;
        ok = ok && hipMalloc((void**)&pp.rlog, log_elems * 3u * sizeof(float) + 16u) == hipSuccess;
        if (compact) ok = ok && hipMalloc((void**)&pp.ovf_slot, (size_t)(paths + 4) * sizeof(uint32_t)) == hipSuccess;
        ok = ok && hipMalloc((void**)&pp.cnt, (size_t)paths * sizeof(uint32_t)) == hipSuccess;
        ok = ok && hipMalloc((void**)&pp.slow_list, (size_t)(paths + 64) * sizeof(uint32_t)) == hipSuccess;
        ok = ok && hipMalloc((void**)&pp.sh_slow_list, (size_t)(paths + 64) * sizeof(uint32_t)) == hipSuccess;
        // ordered on the context's stream: the pipes' streams wait for it before their first launch (fork_pipes)
        ok = ok && hipMemsetAsync(pp.cnt, 0, (size_t)paths * sizeof(uint32_t), ctx->stream) == hipSuccess;
        pp.cur_slots = 0;
        pp.chunk_base = 0;
        pp.chunk_count = 0;
        pp.shadow_pending = false;
    }
    if (!ok)
    {
        free_path_buffers(f);
        (void)hipGetLastError();      // the failed hipMalloc is handled here: do not let it surface at the next launch check
        return fail(ctx, "out of device memory for the per-path buffers");
    }
    return RT_OK;
}

// Cross-stream ordering.  fork: the pipes' own streams wait for everything enqueued on the context's stream
// so far; join: the context's stream waits for everything the pipes have been given.
int fork_pipes(rt_frame* f)
{
    rt_ctx* ctx = f->ctx;
    if (f->n_pipes <= 1) return RT_OK;
    HIPCHK(ctx, hipEventRecord(f->ps[0].done, ctx->stream));
    for (uint32_t i = 1; i < f->n_pipes; ++i) HIPCHK(ctx, hipStreamWaitEvent(f->ps[i].stream, f->ps[0].done, 0));
    return RT_OK;
}

int join_pipes(rt_frame* f)
{
    rt_ctx* ctx = f->ctx;
    for (uint32_t i = 1; i < RT_MAX_PIPES; ++i)
    {
        if (!f->ps[i].stream) continue;
        HIPCHK(ctx, hipEventRecord(f->ps[i].done, f->ps[i].stream));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, f->ps[i].done, 0));
    }
    return RT_OK;
}

// stream, event, counters and spill area of pipes 0 .. count-1 (created once, kept for the frame's life)
int ensure_pipe_resources(rt_frame* f, uint32_t count)
{
    rt_ctx* ctx = f->ctx;
    if (count > RT_MAX_PIPES) count = RT_MAX_PIPES;
    // worst case over the kernel variants: 32 one-wave blocks per CU, 8-entry LDS stack
    const size_t spill_bytes = (size_t)ctx->prop.multiProcessorCount * 32 * 64 * (RT_W4_STACK_MAX - 8) * sizeof(uint2);
    for (uint32_t i = 0; i < (count ? count : 1u); ++i)
    {
        PathPipe& q = f->ps[i];
        if (!q.stream)
        {
            if (i == 0) q.stream = ctx->stream;
            else HIPCHK(ctx, hipStreamCreateWithFlags(&q.stream, hipStreamNonBlocking));
        }
        if (!q.done) HIPCHK(ctx, hipEventCreateWithFlags(&q.done, hipEventDisableTiming));
        if (!q.side)
        {
            // lower priority than the main stream: when both have workgroups waiting, the closest-hit trace and
            // k_shade (the critical chain of a bounce) get the free slots first
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            HIPCHK(ctx, hipStreamCreateWithPriority(&q.side, hipStreamNonBlocking, lo));
            HIPCHK(ctx, hipEventCreateWithFlags(&q.ev_shaded, hipEventDisableTiming));
            for (hipEvent_t& e : q.ev_shadow) HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        if (!q.counters)
        {
            HIPCHK(ctx, hipMalloc((void**)&q.counters, sizeof(DCounters)));
            HIPCHK(ctx, hipMemsetAsync(q.counters, 0, sizeof(DCounters), ctx->stream));
        }
        if (!q.spill) HIPCHK(ctx, hipMalloc((void**)&q.spill, spill_bytes));
        if (!q.sh_spill) HIPCHK(ctx, hipMalloc((void**)&q.sh_spill, spill_bytes));
    }
    return RT_OK;
}

int flush_log(rt_frame* f, bool keep_open = false);
int ensure_whole_tile(rt_frame* f);

// The main stream of the current pipe waits for the shadow trace that last used shadow queue `q` (rt_integrate runs
// those on PathPipe::side); nothing to do when none is outstanding.
int wait_shadow(rt_frame* f, uint32_t q)
{
    if (!f->p->shadow_in_flight[q]) return RT_OK;
    HIPCHK(f->ctx, hipStreamWaitEvent(f->p->stream, f->p->ev_shadow[q], 0));
    f->p->shadow_in_flight[q] = false;
    return RT_OK;
}

// Does the shadow trace of this bounce go to the pipe's side stream?  Inside rt_integrate AND through the stage API (the
// reference's own call pattern, one Integrate() per frame: its launches are small, and the shadow trace of bounce b beside
// the closest-hit trace of bounce b + 1 fills the machine twice as well).  Everything that reads what a shadow trace writes
// waits for it: k_shade(b + 2) (wait_shadow), the log replay (flush_log), rt_reset, the debug readers.
bool side_on(const rt_frame* f) { return f->overlap_shadow != 0 && !(f->denoiser || f->aov != 0); }

// Grows the per-path buffers to hold `want` samples in flight (clamped to slot_cap); they
// are sized by the largest batch actually requested, not by the cap.
int ensure_slots(rt_frame* f, uint32_t want)
{
    uint32_t cap = slot_cap(f);
    if (want > cap) want = cap;
    if (want <= f->slots && 2u * (f->max_bounces + 1u) <= f->log_entries && f->chunk_pixels == chunk_for(f, f->slots) &&
        (f->log_ovf_blocks != 0u) == log_is_compact(f, f->slots))
        return RT_OK;
    if (flush_log(f) != RT_OK || join_pipes(f) != RT_OK) return RT_ERROR;
    HIPCHK(f->ctx, hipStreamSynchronize(f->ctx->stream));
    uint32_t keep = f->slots < cap ? f->slots : cap;
    uint32_t n = want > keep ? want : keep;
    // HBM shared with other frames / processes: halve the batch until the buffers fit
    while (alloc_path_buffers(f, n) != RT_OK)
    {
        if (n == 1) return RT_ERROR;
        n = n > 2u * keep && keep > 0 ? n / 2u : (n > keep ? keep : n / 2u);
        if (n == 0) n = 1;
        f->slots_limit = n;
    }
    return RT_OK;
}

// the radiance log of the current pipe as the kernels see it
DLog dlog(const rt_frame* f)
{
    DLog L;
    L.rlog = f->p->rlog; L.cnt = f->p->cnt; L.ovf_slot = f->log_ovf_blocks ? f->p->ovf_slot : nullptr;
    L.stride = f->log_stride; L.inline_entries = f->log_inline; L.ovf_blocks = f->log_ovf_blocks;
    return L;
}

// The stage API and the per-frame features (AOVs, denoiser) trace ONE sample of the WHOLE tile.  The buffers may have
// been cut into chunks for an earlier, larger batch under RT_OPT_PATH_STATE_LIMIT_MB (ensure_slots keeps an allocation
// that is large enough in SLOTS): re-allocate for one sample in flight, which the caller has checked does fit.
int ensure_whole_tile(rt_frame* f)
{
    if (f->chunk_pixels >= (f->n_local ? f->n_local : 1u)) return RT_OK;
    if (f->p->cur_slots != 0) return fail(f->ctx, "rt_generate_rays: the previous sample was not advanced (rt_advance_sample)");
    if (flush_log(f) != RT_OK || join_pipes(f) != RT_OK) return RT_ERROR;
    HIPCHK(f->ctx, hipStreamSynchronize(f->ctx->stream));
    return alloc_path_buffers(f, 1);
}

// Adds the logged contributions of the batch in flight to the running sum.
int flush_log(rt_frame* f, bool keep_open)
{
    rt_ctx* ctx = f->ctx;
    if (f->p->cur_slots == 0 || f->n_local == 0 || f->p->chunk_count == 0) { f->p->cur_slots = 0; return RT_OK; }
    if (f->p->shadow_pending)
        return fail(ctx, "radiance requested between rt_shade and rt_intersect_shadow (direct samples still tentative)");
    if (wait_shadow(f, 0) != RT_OK || wait_shadow(f, 1) != RT_OK) return RT_ERROR;   // their verdicts are in the log
    uint32_t blocks = (f->p->chunk_count + 255u) / 256u;
    hipLaunchKernelGGL(k_flush, dim3(blocks), dim3(256), 0, f->p->stream, f->radiance + f->p->chunk_base, dlog(f),
        f->p->chunk_count, f->p->cur_slots, f->chunk_pixels, keep_open && f->log_ovf_blocks != 0u ? 1u : 0u, 0u);
    if (hipGetLastError() != hipSuccess) return fail(ctx, "k_flush launch failed");
    f->p->cur_slots = 0;
    return RT_OK;
}

} // namespace

// waits (host side) for every stream a frame launches on besides the context's own
static int sync_frame_streams(rt_frame* f)
{
    rt_ctx* ctx = f->ctx;
    for (PathPipe& q : f->ps)
    {
        if (q.stream && q.stream != ctx->stream) HIPCHK(ctx, hipStreamSynchronize(q.stream));
        if (q.side) HIPCHK(ctx, hipStreamSynchronize(q.side));
    }
    if (f->present_stream) HIPCHK(ctx, hipStreamSynchronize(f->present_stream));   // an image on its way to the host (rt_frame_present)
    f->present_pending = false;
    return RT_OK;
}

extern "C" { static int deferred_materialize(rt_frame* f); }   // RT_OPT_FRAME_KERNEL (below, with the stage functions)

namespace
{
// Mid-sample read (stage API / debugging): apply what has been logged so far but
// keep the sample open -- later contributions append again from entry 0.
int flush_log_keep(rt_frame* f)
{
    uint32_t keep = f->p->cur_slots;
    int rc = flush_log(f, true);
    f->p->cur_slots = keep;
    return rc;
}

// The stage API over every pipe that holds a chunk of its sample (RT_OPT_STAGE_PIPES; one pipe is the plain case): `body` runs with f->p on
// each of them in turn.
template <class F>
int for_stage_pipes(rt_frame* f, F&& body)
{
    int rc = RT_OK;
    const uint32_t n = f->stage_chunks ? f->stage_chunks : 1u;
    for (uint32_t i = 0; i < n && rc == RT_OK; ++i)
    {
        f->p = &f->ps[i];
        rc = body();
    }
    f->p = &f->ps[0];
    return rc;
}

// ... their logs replayed into the radiance (keep: the sample stays open), and the context's stream made to see all of them
int flush_stage(rt_frame* f, bool keep = false)
{
    if (f->deferred.active && deferred_materialize(f) != RT_OK) return RT_ERROR;    // RT_OPT_FRAME_KERNEL: the recorded stages run after all
    const uint32_t n = f->stage_chunks ? f->stage_chunks : 1u;
    if (for_stage_pipes(f, [&]() { return keep ? flush_log_keep(f) : flush_log(f); }) != RT_OK) return RT_ERROR;
    return n > 1 ? join_pipes(f) : RT_OK;
}
} // namespace

extern "C" {

static int create_frame(rt_ctx* ctx, const rt_frame_desc* fd, rt_frame** out, hipStream_t borrowed_main_stream);

int rt_frame_create(rt_ctx* ctx, const rt_frame_desc* fd, rt_frame** out) { return create_frame(ctx, fd, out, nullptr); }

// borrowed_main_stream != NULL: pipe 0 launches there instead of on the context's stream (a bank of RT_OPT_SAMPLES_AHEAD; the stream stays the lender's)
static int create_frame(rt_ctx* ctx, const rt_frame_desc* fd, rt_frame** out, hipStream_t borrowed_main_stream)
{
    if (!ctx || !fd || !out) return fail(ctx, "rt_frame_create: NULL argument");
    *out = nullptr;
    if (fd->width == 0 || fd->height == 0) return fail(ctx, "rt_frame_create: empty image");
    if (fd->tile_count == 0 || fd->tile_rank >= fd->tile_count || fd->band_height == 0)
        return fail(ctx, "rt_frame_create: bad tile description");
    (void)hipSetDevice(ctx->device);
    rt_frame* f = new rt_frame();
    f->ctx = ctx;
    f->tile.width = fd->width; f->tile.height = fd->height;
    f->tile.band_h = fd->band_height; f->tile.rank = fd->tile_rank; f->tile.nranks = fd->tile_count;
    uint32_t rows = 0;
    for (uint32_t band = fd->tile_rank; (uint64_t)band * fd->band_height < fd->height; band += fd->tile_count)
    {
        uint32_t start = band * fd->band_height;
        uint32_t h = fd->height - start < fd->band_height ? fd->height - start : fd->band_height;
        rows += h;
    }
    f->tile.local_rows = rows;
    if ((uint64_t)rows * fd->width > 0x7FFFFFFFull) { delete f; return fail(ctx, "rt_frame_create: tile too large"); }
    f->n_local = rows * fd->width;
    size_t n = f->n_local ? f->n_local : 1;
    f->trace_blocks = (uint32_t)ctx->prop.multiProcessorCount * 12u;
    f->trace_blocks = (f->trace_blocks + 7u) & ~7u;
    f->radiance = nullptr;
    f->resolved = nullptr;
    f->ps[0].stream = borrowed_main_stream;              // (NULL: ensure_pipe_resources gives pipe 0 the context's stream)
    bool ok = true;
    ok = ok && hipMalloc((void**)&f->radiance, n * sizeof(float4)) == hipSuccess;
    ok = ok && hipMalloc((void**)&f->resolved, n * sizeof(float4)) == hipSuccess;
    ok = ok && hipMalloc((void**)&f->aov_buf.diffuse_albedo, n * sizeof(float4)) == hipSuccess;
    ok = ok && hipMalloc((void**)&f->aov_buf.depth, n * sizeof(float)) == hipSuccess;
    ok = ok && hipMalloc((void**)&f->aov_buf.normal, n * sizeof(float4)) == hipSuccess;
    ok = ok && hipMalloc((void**)&f->aov_buf.velocity, n * sizeof(float2)) == hipSuccess;
    ok = ok && hipMalloc((void**)&f->prev_radiance, n * sizeof(float4)) == hipSuccess;
    ok = ok && hipMalloc((void**)&f->prev_depth, n * sizeof(float)) == hipSuccess;
    ok = ok && hipMemsetAsync(f->aov_buf.diffuse_albedo, 0, n * sizeof(float4), ctx->stream) == hipSuccess;
    ok = ok && hipMemsetAsync(f->aov_buf.depth, 0, n * sizeof(float), ctx->stream) == hipSuccess;
    ok = ok && hipMemsetAsync(f->aov_buf.normal, 0, n * sizeof(float4), ctx->stream) == hipSuccess;
    ok = ok && hipMemsetAsync(f->aov_buf.velocity, 0, n * sizeof(float2), ctx->stream) == hipSuccess;
    ok = ok && hipMemsetAsync(f->prev_radiance, 0, n * sizeof(float4), ctx->stream) == hipSuccess;
    ok = ok && hipMemsetAsync(f->prev_depth, 0, n * sizeof(float), ctx->stream) == hipSuccess;
    ok = ok && alloc_path_buffers(f, 1) == RT_OK;     // grows on demand (ensure_slots)
    ok = ok && ensure_pipe_resources(f, f->pipelines) == RT_OK;
    if (!ok)
    {
        rt_frame_destroy(f);
        (void)hipGetLastError();
        return fail(ctx, "rt_frame_create: out of device memory");
    }
    memset(&f->camera, 0, sizeof(f->camera));
    memset(&f->camera_last, 0, sizeof(f->camera_last));
    memset(&f->prev_camera, 0, sizeof(f->prev_camera));
    *out = f;
    ctx->frames.push_back(f);
    return rt_reset(f);                                  // the reference ctor ends with Reset(), cl_pt_integrator.cpp:258
}

int rt_frame_destroy(rt_frame* f)
{
    if (!f) return RT_OK;
    (void)hipSetDevice(f->ctx->device);
    ahead_destroy(f);                                     // its banks first (frames of their own)
    f->ctx->frames.erase(std::remove(f->ctx->frames.begin(), f->ctx->frames.end(), f), f->ctx->frames.end());
    for (PathPipe& q : f->ps)
    {
        if (q.stream) (void)hipStreamSynchronize(q.stream);
        if (q.side) (void)hipStreamSynchronize(q.side);
    }
    (void)hipStreamSynchronize(f->ctx->stream);
    free_path_buffers(f);
    void* ptrs[] = {f->radiance, f->resolved, f->aov_buf.diffuse_albedo, f->aov_buf.depth,
        f->aov_buf.normal, f->aov_buf.velocity, f->prev_radiance, f->prev_depth};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (uint32_t i = 0; i < RT_MAX_PIPES; ++i)
    {
        PathPipe& q = f->ps[i];
        if (q.counters) (void)hipFree(q.counters);
        if (q.spill) (void)hipFree(q.spill);
        if (q.sh_spill) (void)hipFree(q.sh_spill);
        if (q.done) (void)hipEventDestroy(q.done);
        if (q.side) (void)hipStreamSynchronize(q.side);
        if (q.ev_shaded) (void)hipEventDestroy(q.ev_shaded);
        for (hipEvent_t e : q.ev_shadow) if (e) (void)hipEventDestroy(e);
        if (q.side) (void)hipStreamDestroy(q.side);
        if (i > 0 && q.stream) (void)hipStreamDestroy(q.stream);
    }
    if (f->present_stream) { (void)hipStreamSynchronize(f->present_stream); (void)hipStreamDestroy(f->present_stream); }
    if (f->frame_counts) (void)hipFree(f->frame_counts);
    if (f->frame_slow) (void)hipFree(f->frame_slow);
    if (f->frame_spill) (void)hipFree(f->frame_spill);
    for (auto& pair : f->fk_auto.ev) for (hipEvent_t e : pair) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : f->ev_resolved) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : f->ev_copied) if (e) (void)hipEventDestroy(e);
    if (f->resolved_b) (void)hipFree(f->resolved_b);
    for (auto& s : f->spans) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
    for (auto e : f->event_pool) (void)hipEventDestroy(e);
    delete f;
    return RT_OK;
}

uint32_t rt_frame_local_rows(rt_frame* f) { return f ? f->tile.local_rows : 0; }

uint32_t rt_frame_global_row(rt_frame* f, uint32_t ly)
{
    if (!f) return 0;
    uint32_t band = ly / f->tile.band_h;
    return (band * f->tile.nranks + f->tile.rank) * f->tile.band_h + (ly - band * f->tile.band_h);
}

int rt_set_option(rt_frame* f, int option, uint32_t value)
{
    if (!f) return fail(nullptr, "rt_set_option: frame is NULL");
    if (f->deferred.active && option != RT_OPT_FRAME_KERNEL && deferred_materialize(f) != RT_OK) return RT_ERROR;   // an option changed between two recorded stages
    if (f->ahead && option != RT_OPT_PROFILE_KERNELS && !(option == RT_OPT_MAX_BOUNCES && value == f->max_bounces) && !(option == RT_OPT_WHITE_FURNACE && (value ? 1u : 0u) == f->white_furnace))
        ahead_discard(f);                                 // (the two exceptions: HIPPathTraceIntegrator::SyncOptions sets them before every frame)
    switch (option)
    {
    case RT_OPT_SAMPLES_AHEAD:
        if ((value & 0xFFu) > 64u && (value & 0xFFu) != 255u) return fail(f->ctx, "rt_set_option: RT_OPT_SAMPLES_AHEAD is 0 (off), 1 (automatic depth) or 2..64 samples per batch (+ 256: one stream per bank)");
        if (f->ahead_owner) return fail(f->ctx, "rt_set_option: RT_OPT_SAMPLES_AHEAD on a bank");
        f->ahead_opt = value & 0x1FFu;
        if (f->ahead) f->ahead->configured = false;
        return RT_OK;
    case RT_OPT_MAX_BOUNCES:
        if (value > RT_MAX_BOUNCES_LIMIT) return fail(f->ctx, "rt_set_option: max_bounces above RT_MAX_BOUNCES_LIMIT");
        if (value != f->max_bounces)
        {
            f->fk_auto.frames = 0; f->fk_auto.timing = -1; f->fk_auto.decided = false; f->fk_auto.use_kernel = false;   //  another workload: RT_OPT_FRAME_KERNEL = 255 measures again
            if (flush_log(f) != RT_OK) return RT_ERROR;
            HIPCHK(f->ctx, hipStreamSynchronize(f->ctx->stream));
            f->max_bounces = value;
            f->log_full_forced = false;               // another path length: the compact layout gets another chance
            f->fallback_limit_mb = 0;
            return ensure_slots(f, 1);                // log rows / payload split follow max_bounces
        }
        return RT_OK;
    case RT_OPT_SAMPLES_IN_FLIGHT:
        if (value > 1024) return fail(f->ctx, "rt_set_option: samples in flight must be 0 (auto) or 1..1024");
        if (value != f->slots_opt)
        {
            if (flush_log(f) != RT_OK) return RT_ERROR;
            HIPCHK(f->ctx, hipStreamSynchronize(f->ctx->stream));
            uint32_t old = f->slots_opt, old_slots = f->slots;
            f->slots_opt = value;
            f->reserved_explicitly = false; f->samples_asked = 0;      // (the caller decides anew how the buffers are sized: lean growth starts over)
            // an explicit count is allocated now; auto (0) grows with the batches requested
            if (alloc_path_buffers(f, value ? value : 1u) != RT_OK)
            {
                f->slots_opt = old;
                (void)alloc_path_buffers(f, old_slots);
                return RT_ERROR;
            }
        }
        return RT_OK;
    case RT_OPT_WHITE_FURNACE: f->white_furnace = value ? 1 : 0; return RT_OK;
    case RT_OPT_SAMPLER:
        if (value > 1) return fail(f->ctx, "rt_set_option: sampler must be 0 (kRandom) or 1 (kBlueNoise)");
        if (value == 1 && !f->ctx->blue_noise)
            return fail(f->ctx, "rt_set_option: SamplerType::kBlueNoise needs rt_upload_blue_noise_tables first");
        if (value != f->sampler) { f->fk_auto.frames = 0; f->fk_auto.timing = -1; f->fk_auto.decided = false; f->fk_auto.use_kernel = false; }
        f->sampler = value;
        return RT_OK;
    case RT_OPT_AOV:
        if (value > 4) return fail(f->ctx, "rt_set_option: AOV index must be 0..4");
        f->aov = value;
        return RT_OK;
    case RT_OPT_DENOISER:
        // 1: the frame runs TemporalAccumulation itself -- the reprojection crosses rows, so it needs the whole image;
        // 2: the frame only prepares the filter's inputs (reset every frame, one sample, depth and motion AOVs) and
        //    rt_group_denoise runs the filter on the gathered image: the mode for tiles.
        if (value > 2) return fail(f->ctx, "rt_set_option: RT_OPT_DENOISER is 0, 1 or 2");
        if (value == 1 && f->tile.nranks != 1)
            return fail(f->ctx, "rt_set_option: the temporal denoiser reprojects across rows and needs the whole image "
                                "on one GPU (tile_count == 1); on tiles use RT_OPT_DENOISER = 2 + rt_group_denoise");
        f->denoiser = value;
        return RT_OK;
    case RT_OPT_TRACE_DROP_LAST_BOUNCE_RAYS:
        if ((value ? 1u : 0u) != f->drop_last) { f->fk_auto.frames = 0; f->fk_auto.timing = -1; f->fk_auto.decided = false; f->fk_auto.use_kernel = false; }
        f->drop_last = value ? 1 : 0;
        return RT_OK;
    case RT_OPT_PROFILE_KERNELS: f->profile = value ? 1 : 0; return RT_OK;
    case RT_OPT_TRACE_WAVES_PER_CU: f->trace_waves_per_cu = value; return RT_OK;
    case RT_OPT_TRACE_PACKET_BOUNCES:
        if (value != 0) return fail(f->ctx, "rt_set_option: the packet kernel was removed (RT_OPT_TRACE_PACKET_BOUNCES accepts only 0)");
        return RT_OK;
    case RT_OPT_TRACE_SELECT_FORM_BOX: f->select_form_box = value ? RT_SIGN_SLOW : 0u; return RT_OK;
    case RT_OPT_TRACE_TUNE: f->trace_tune = value; return RT_OK;
    case RT_OPT_SHADE_PARTITION: f->shade_partition = value & 3u; return RT_OK;
    case RT_OPT_OVERLAP_SHADOW: f->overlap_shadow = value ? 1u : 0u; return RT_OK;
    case RT_OPT_PIPELINES:
        if (value == 0 || value > RT_MAX_PIPES) return fail(f->ctx, "rt_set_option: pipelines must be 1..RT_MAX_PIPES");
        if (value != f->pipelines)
        {
            if (flush_log(f) != RT_OK || join_pipes(f) != RT_OK) return RT_ERROR;
            HIPCHK(f->ctx, hipStreamSynchronize(f->ctx->stream));
            if (ensure_pipe_resources(f, value) != RT_OK) return RT_ERROR;
            f->pipelines = value;
            return alloc_path_buffers(f, f->slots);
        }
        return RT_OK;
    case RT_OPT_FRAME_KERNEL:
        if (f->deferred.active) return fail(f->ctx, "rt_set_option: RT_OPT_FRAME_KERNEL cannot change while a sample is in flight (rt_advance_sample first)");
        if (value != f->frame_kernel)
        {
            // (the grid and its buffers are sized by the value: the next launch makes them again)
            HIPCHK(f->ctx, hipStreamSynchronize(f->ctx->stream));
            if (f->frame_counts) (void)hipFree(f->frame_counts);
            if (f->frame_slow) (void)hipFree(f->frame_slow);
            if (f->frame_spill) (void)hipFree(f->frame_spill);
            f->frame_counts = nullptr; f->frame_slow = nullptr; f->frame_spill = nullptr; f->frame_blocks = 0;
            f->frame_kernel = value == 255u ? 255u : (value > 64u ? 64u : value);
            f->fk_auto.frames = 0; f->fk_auto.timing = -1; f->fk_auto.decided = false; f->fk_auto.use_kernel = false;
        }
        return RT_OK;
    case RT_OPT_STAGE_PIPES:
        if (value == 0 || value > RT_MAX_PIPES) return fail(f->ctx, "rt_set_option: stage pipes must be 1..RT_MAX_PIPES");
        if (value != f->stage_pipes)
        {
            if (f->p->cur_slots != 0) return fail(f->ctx, "rt_set_option: RT_OPT_STAGE_PIPES cannot change while a sample is in flight (rt_advance_sample first)");
            if (flush_log(f) != RT_OK || join_pipes(f) != RT_OK) return RT_ERROR;
            HIPCHK(f->ctx, hipStreamSynchronize(f->ctx->stream));
            if (ensure_pipe_resources(f, value) != RT_OK) return RT_ERROR;
            f->stage_pipes = value;
            f->stage_chunks = 1;
            if (f->slots == 1) return alloc_path_buffers(f, 1);
        }
        return RT_OK;
    case RT_OPT_PATH_STATE_LIMIT_MB:
        if (value != f->state_limit_mb)
        {
            if (flush_log(f) != RT_OK) return RT_ERROR;
            HIPCHK(f->ctx, hipStreamSynchronize(f->ctx->stream));
            uint32_t old = f->state_limit_mb;
            f->state_limit_mb = value;
            if (alloc_path_buffers(f, f->slots) != RT_OK)
            {
                f->state_limit_mb = old;
                (void)alloc_path_buffers(f, f->slots);
                return RT_ERROR;
            }
        }
        return RT_OK;
    case RT_OPT_DEBUG_ALLOC_LIMIT: f->debug_alloc_limit = value; return RT_OK;
    case RT_OPT_SMALL_LAUNCH_PATHS: f->small_launch_paths = value; f->small_launch_set = true; return RT_OK;
    case RT_OPT_TRACE_TAIL_LANES: f->trace_tail_lanes = value > 64u ? 64u : value; return RT_OK;
    case RT_OPT_TRACE_TAIL_PATHS: f->trace_tail_paths = value; return RT_OK;
    case RT_OPT_CHUNK_REFILL: f->chunk_refill = value ? 1u : 0u; return RT_OK;
    case RT_OPT_COMPACT_LOG:
    case RT_OPT_DEBUG_LOG_POOL_DIV:
        if (option == RT_OPT_DEBUG_LOG_POOL_DIV && value == 0) return fail(f->ctx, "rt_set_option: RT_OPT_DEBUG_LOG_POOL_DIV must be >= 1");
        if (flush_log(f) != RT_OK) return RT_ERROR;
        HIPCHK(f->ctx, hipStreamSynchronize(f->ctx->stream));
        if (option == RT_OPT_COMPACT_LOG && value > 2u) return fail(f->ctx, "rt_set_option: RT_OPT_COMPACT_LOG is 0, 1 or 2");
        if (option == RT_OPT_COMPACT_LOG) f->compact_log_opt = value; else f->log_pool_div = value;
        f->log_full_forced = false;
        f->fallback_limit_mb = 0;
        return alloc_path_buffers(f, f->slots);
    case RT_OPT_TRACE_VARIANT:
        if (!(value == 0 || value == 5 || (value >= 8 && value <= 11))) return fail(f->ctx, "rt_set_option: unknown trace kernel variant (0, 5, 8..11)");
        f->trace_variant = value;
        return RT_OK;
    default: return fail(f->ctx, "rt_set_option: unknown option");
    }
}

int rt_set_camera(rt_frame* f, const rt_camera* camera)
{
    if (!f || !camera) return fail(nullptr, "rt_set_camera: NULL argument");
    //  a camera set between rt_generate_rays and rt_advance_sample: the recorded stages belong to the OLD camera -- they run now, with it
    if (f->deferred.active && memcmp(&f->camera, camera, sizeof(rt_camera)) != 0 && deferred_materialize(f) != RT_OK) return RT_ERROR;
    if (f->ahead && memcmp(&f->camera, camera, sizeof(rt_camera)) != 0) ahead_discard(f);              // whatever was traced ahead (and the quiet count) belonged to another view
                                                                                                       // (HIPPathTraceIntegrator sets the SAME camera before every frame: nothing happens)
    f->camera = *camera;
    f->prev_camera = f->camera_last;     // what GenerateAOV sees as prev_camera this frame
    f->camera_last = *camera;
    return RT_OK;
}

// ---- stages ----------------------------------------------------------------
namespace
{
// RAII bracket: records a start event now and a stop event at scope exit
struct KernelSpan
{
    rt_frame* f; int cls; hipStream_t st; hipEvent_t a = nullptr, b = nullptr;
    KernelSpan(rt_frame* f_, int cls_, hipStream_t stream = nullptr) : f(f_), cls(cls_), st(stream ? stream : f_->p->stream)
    {
        if (!f->profile) return;
        auto get = [&]() { hipEvent_t e = nullptr; if (!f->event_pool.empty()) { e = f->event_pool.back(); f->event_pool.pop_back(); } else (void)hipEventCreate(&e); return e; };
        a = get(); b = get();
        (void)hipEventRecord(a, st);
    }
    ~KernelSpan()
    {
        if (!a) return;
        (void)hipEventRecord(b, st);
        f->spans.push_back({a, b, cls});
    }
};
} // namespace

#define FRAME_PROLOGUE(f, name)                                                         \
    if (!(f)) return fail(nullptr, name ": frame is NULL");                             \
    rt_ctx* ctx = (f)->ctx;                                                             \
    if (!ctx->scene.valid) return fail(ctx, name ": no scene uploaded");                \
    (void)hipSetDevice(ctx->device)

} // extern "C"

namespace
{
// k_trace2: separate wave-uniform loops (trace_kernels.h).  One-wave blocks; the persistent grid is sized to the
// LDS-limited residency: 160 KiB / (entries * 512 B) per CU.
#define RT_TRACE2_DEFAULT_TUNE (32u | (8u << 8))     // profiles/r02_ktrace2_tune_sweep.log, r02_w4_tune_sweep.log
#define RT_W4_DEFAULT_RAYS_PER_LANE 0u               // k_trace_w4's live-counter grid (trace_kernels.h); 0 = every wave
template <bool SHADOW, int STACK>
void launch_trace2(rt_frame* f, const float4* o4, const float4* d4, const uint32_t* aux, const uint32_t* count)
{
    rt_ctx* ctx = f->ctx;
    uint32_t per_cu = (160u * 1024u) / (STACK * 512u);
    if (per_cu > 32u) per_cu = 32u;
    if (f->trace_waves_per_cu && f->trace_waves_per_cu < per_cu) per_cu = f->trace_waves_per_cu;
    uint32_t blocks = ((uint32_t)ctx->prop.multiProcessorCount * per_cu + 7u) & ~7u;
    uint32_t tune = f->trace_tune ? f->trace_tune : RT_TRACE2_DEFAULT_TUNE;
    if ((tune & 0xFFu) > 64u) tune = (tune & ~0xFFu) | 64u;
    if ((tune & 0xFFu) == 0u) tune |= 1u;
    hipLaunchKernelGGL((k_trace2<SHADOW, STACK>), dim3(blocks), dim3(64), 0, f->tl_stream, ctx->scene.d, o4, d4, aux, count,
        &f->p->counters->head[f->tl_flavour][0], SHADOW ? (float4*)nullptr : f->p->hits,
        dlog(f), f->select_form_box, f->tl_spill, tune, (const uint32_t*)nullptr,
        &f->p->counters->stack_spills);
}

// k_trace_w4 over the 4-wide quantized tree, then k_trace2 over the (normally empty) list of rays it left out
template <bool SHADOW, int STACK>
void launch_trace_w4(rt_frame* f, const float4* o4, const float4* d4, const uint32_t* aux, const uint32_t* count, uint32_t /*bounce*/)
{
    rt_ctx* ctx = f->ctx;
    uint32_t per_cu = (160u * 1024u) / (STACK * 512u);
    if (per_cu > 32u) per_cu = 32u;
    if (f->trace_waves_per_cu && f->trace_waves_per_cu < per_cu) per_cu = f->trace_waves_per_cu;
    uint32_t blocks = ((uint32_t)ctx->prop.multiProcessorCount * per_cu + 7u) & ~7u;
    // RT_OPT_TRACE_TUNE, field by field (0 = that field's default)
    const uint32_t t = f->trace_tune;
    uint32_t node_q = (t & 0xFFu) ? (t & 0xFFu) : (RT_TRACE2_DEFAULT_TUNE & 0xFFu);
    const uint32_t leaf_q = ((t >> 8) & 0xFFu) ? ((t >> 8) & 0xFFu) : ((RT_TRACE2_DEFAULT_TUNE >> 8) & 0xFFu);
    if (node_q > 64u) node_q = 64u;
    uint32_t rays_per_lane = (t >> 24) ? (t >> 24) : RT_W4_DEFAULT_RAYS_PER_LANE;
    if (rays_per_lane == 255u) rays_per_lane = 0u;                          // 255 = every wave of the residency-sized grid
    const uint32_t tune = node_q | leaf_q << 8 | (t & 0xFF0000u) | rays_per_lane << 24;
    const uint32_t s = f->tl_flavour;
    // the instance with loop D and refilled chunks (below): static assignment stays ahead of the shared work heads up to larger launches
    // there -- 8 M rays instead of 3 M (4 / 8 / 16 samples of a 1080p frame in flight: +2.7 / +8.4 / +4.8 %, profiles/r04_call20.log)
    const uint64_t batch_paths = (uint64_t)f->p->chunk_count * (f->p->cur_slots ? f->p->cur_slots : 1u);
    const bool tail_instance = STACK == 12 && f->trace_tail_lanes != 0u && batch_paths < f->trace_tail_paths && !(!SHADOW && f->timeline);
    const uint64_t small = f->small_launch_set || !tail_instance ? f->small_launch_paths : 8000000ull;
    const uint32_t chunk_below = small > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)small;
    // (Camera-ray launches refilled instead of chunked -- coherent rays, short tails? -- measured: 2477 instead of 2876 Mrays/s per frame,
    // profiles/r04_call14.log: a refilling launch of any size pays its ~0.6 ms drain.)
    unsigned long long* const no_timeline = nullptr;
    // the instance with loop D (the fused tail pass) where the whole batch is a small launch: the kernel then runs in chunk
    // mode whatever its live counter says (count <= paths < chunk_below)
    const bool tail = tail_instance;
    if (!SHADOW && STACK == 12 && f->timeline)          // tools/launch_timeline.py: the instrumented instance
        hipLaunchKernelGGL((k_trace_w4<false, 12, true, false>), dim3(blocks), dim3(64), 0, f->tl_stream, ctx->scene.d, o4, d4, aux, count,
            &f->p->counters->head[s][0], f->p->hits, dlog(f), f->tl_spill, tune, f->tl_slow_list,
            &f->p->counters->slow_count[s], &f->p->counters->stack_spills, &f->p->counters->tl_start[f->timeline_bounce & 63u],
            f->timeline_bounce & 63u, chunk_below, 0u, f->chunk_refill);
    else if (tail)
        hipLaunchKernelGGL((k_trace_w4<SHADOW, 12, false, true>), dim3(blocks), dim3(64), 0, f->tl_stream, ctx->scene.d, o4, d4, aux, count,
            &f->p->counters->head[s][0], SHADOW ? (float4*)nullptr : f->p->hits, dlog(f),
            f->tl_spill, tune, f->tl_slow_list, &f->p->counters->slow_count[s], &f->p->counters->stack_spills, no_timeline, 0u, chunk_below, f->trace_tail_lanes, f->chunk_refill);
    else
        hipLaunchKernelGGL((k_trace_w4<SHADOW, STACK, false, false>), dim3(blocks), dim3(64), 0, f->tl_stream, ctx->scene.d, o4, d4, aux, count,
            &f->p->counters->head[s][0], SHADOW ? (float4*)nullptr : f->p->hits, dlog(f),
            f->tl_spill, tune, f->tl_slow_list, &f->p->counters->slow_count[s], &f->p->counters->stack_spills, no_timeline, 0u, chunk_below, 0u, f->chunk_refill);
    // The follow-up over the (normally empty) slow list: waves with a two-entry LDS stack (the rest of the stack
    // lives in the spill area) -- 1 KiB of LDS and a few registers, so it finds room beside the resident waves of the
    // OTHER stream's persistent launch (RT_OPT_OVERLAP_SHADOW) instead of waiting for that launch to end: with the
    // 6 KiB blocks of the ordinary k_trace2 the closest-hit follow-up sat 2.8 ms on average behind the shadow trace
    // (profiles/r02_final_rocprofv3_kernel_stats_overlap.csv) with k_shade queued behind it.
    // Four such waves per CU (4 KiB of LDS beside the other launch's 26 x 6 KiB): with an empty list they leave at once;
    // with a long one (every shadow ray towards an axis-aligned point-light arrangement, say) the list is not serialised
    // onto one wave per CU.  Scenes whose directional lights make MOST shadow rays slow never get here (Scene::slow_shadow).
    uint32_t blocks2 = ((uint32_t)ctx->prop.multiProcessorCount * 4u + 7u) & ~7u;
    hipLaunchKernelGGL((k_trace2<SHADOW, 2>), dim3(blocks2), dim3(64), 0, f->tl_stream, ctx->scene.d, o4, d4, aux,
        (const uint32_t*)&f->p->counters->slow_count[s], &f->p->counters->slow_head[s][0], SHADOW ? (float4*)nullptr : f->p->hits,
        dlog(f), f->select_form_box, f->tl_spill, tune & 0xFFFFu, (const uint32_t*)f->tl_slow_list,
        &f->p->counters->stack_spills);
}

template <bool SHADOW>
void launch_trace(rt_frame* f, const float4* o4, const float4* d4, const uint32_t* aux, const uint32_t* count,
    uint32_t bounce)
{
    rt_ctx* ctx = f->ctx;
    uint32_t variant = f->trace_variant;
    if (variant == 5)
    {
        // auto: the 4-wide quantized tree wherever it was built -- 4214 (round 1's k_trace) / 4400 (k_trace2) / 5042
        // (k_trace_w4) Mrays/s on the headline workload when it was introduced (profiles/r02_w4_tune_sweep.log).
        // Scenes whose directional lights make most shadow rays "slow" (Scene::slow_shadow) trace the shadow queue with
        // k_trace2, which handles such rays inline.
        variant = (SHADOW && ctx->scene.slow_shadow) ? 8u : 10u;
        // k_trace_w4 switches to its chunk mode by itself when a launch is small (RT_OPT_SMALL_LAUNCH_PATHS, decided in the
        // kernel from the live queue counter).  Without the wide tree, small batches take the per-ray loop of v1, which
        // beats the refilling BVH2 kernel there (1511 vs 737 Mrays/s at one 1080p sample in flight, profiles/r03_call02_*)
        const uint64_t paths = (uint64_t)f->p->chunk_count * (f->p->cur_slots ? f->p->cur_slots : 1u);
        if ((variant == 8u || !ctx->scene.wide_ok) && paths < 2000000ull) variant = 0u;
    }
    if ((variant == 10u || variant == 11u) && !ctx->scene.wide_ok) variant = 8u;
    if ((variant == 8u || variant == 9u) && !ctx->scene.offsets32) variant = 0u;
    switch (variant)
    {
    case 0:
        hipLaunchKernelGGL(k_trace_v1<SHADOW>, dim3(f->trace_waves_per_cu ? (((uint32_t)ctx->prop.multiProcessorCount *
            (f->trace_waves_per_cu < 13u ? f->trace_waves_per_cu : 13u) + 7u) & ~7u) : f->trace_blocks), dim3(64), 0, f->tl_stream, ctx->scene.d, o4, d4,
            aux, count, SHADOW ? (float4*)nullptr : f->p->hits, dlog(f), f->select_form_box, f->tl_spill);
        break;
    case 8:
        if (!SHADOW && ctx->scene.d.entry_ref < 4000000u) launch_trace2<SHADOW, 10>(f, o4, d4, aux, count);
        else launch_trace2<SHADOW, 12>(f, o4, d4, aux, count);
        break;
    case 9: launch_trace2<SHADOW, 12>(f, o4, d4, aux, count); break;
    case 11: launch_trace_w4<SHADOW, 16>(f, o4, d4, aux, count, bounce); break;
    default: launch_trace_w4<SHADOW, 12>(f, o4, d4, aux, count, bounce); break;
    }
}
} // namespace

extern "C" {

int rt_reset(rt_frame* f)                               // CLPathTraceIntegrator::Reset, cl_pt_integrator.cpp:497-508
{
    if (!f) return fail(nullptr, "rt_reset: frame is NULL");
    rt_ctx* ctx = f->ctx;
    (void)hipSetDevice(ctx->device);
    if (join_pipes(f) != RT_OK) return RT_ERROR;
    ahead_discard(f);                        // RT_OPT_SAMPLES_AHEAD: the samples traced ahead continued a sum that starts again
    if (!f->denoiser) f->sample_count = 0;   // Reset() keeps the frame index while denoising (:499-504)
    for (uint32_t i = 0; i < RT_MAX_PIPES; ++i)           // a shadow trace still running on a side stream writes the log
    {
        f->p = &f->ps[i];
        if (f->p->stream && (wait_shadow(f, 0) != RT_OK || wait_shadow(f, 1) != RT_OK)) { f->p = &f->ps[0]; return RT_ERROR; }
    }
    f->p = &f->ps[0];
    f->stage_chunks = 1;
    f->deferred.active = false; f->deferred.ahead = false;  // a recorded sample that was never advanced: nothing has run
    for (uint32_t i = 0; i < RT_MAX_PIPES; ++i)
    {
        PathPipe& q = f->ps[i];
        q.prev_bounces = 0;
        q.cur_slots = 0;
        q.chunk_base = 0;
        q.chunk_count = 0;
        q.fold_accumulates = 0;
        q.shadow_pending = false;
        if (q.cnt) HIPCHK(ctx, hipMemsetAsync(q.cnt, 0, (size_t)f->log_stride * sizeof(uint32_t), ctx->stream));
        if (q.counters) HIPCHK(ctx, hipMemsetAsync(q.counters, 0, sizeof(DCounters), ctx->stream));
    }
    HIPCHK(ctx, hipMemsetAsync(f->radiance, 0, (size_t)(f->n_local ? f->n_local : 1) * sizeof(float4), ctx->stream));
    return RT_OK;
}

} // extern "C"

namespace
{
// Primary rays for `n_slots` consecutive samples (sample indices sample_count ..
// sample_count + n_slots - 1) in one launch; they then travel through the same queues.
int generate_rays(rt_frame* f, uint32_t n_slots, uint32_t chunk_base = 0, bool later_chunk_on_this_pipe = false)
{
    rt_ctx* ctx = f->ctx;
    if (f->p->cur_slots != 0) return fail(ctx, "rt_generate_rays: the previous sample was not advanced (rt_advance_sample)");
    if (chunk_base == 0 && ensure_slots(f, n_slots) != RT_OK) return RT_ERROR;
    if (n_slots > f->slots) return fail(ctx, "rt_generate_rays: more samples than the frame can keep in flight");
    if (chunk_base >= (f->n_local ? f->n_local : 1u)) return fail(ctx, "rt_generate_rays: chunk outside the tile");
    f->p->chunk_base = chunk_base;
    f->p->chunk_count = f->n_local - chunk_base < f->chunk_pixels ? f->n_local - chunk_base : f->chunk_pixels;
    float tan_half_fov = rt_tanf(0.5f * f->camera.fov);  // raygeneration.cl:108, uniform -> host
    uint32_t blocks = (f->p->chunk_count * n_slots + 255u) / 256u;
    if (blocks == 0) blocks = 1;
    KernelSpan span(f, 0);
    hipLaunchKernelGGL(k_raygen, dim3(blocks), dim3(256), 0, f->p->stream, f->tile, f->camera, f->sample_count, n_slots,
        tan_half_fov, f->p->prev_bounces, f->p->o4[0], f->p->d4[0], f->p->thr[0], f->p->counters, f->p->chunk_base, f->p->chunk_count,
        f->chunk_pixels, f->p->fold_accumulates);
    f->p->fold_accumulates = later_chunk_on_this_pipe ? 1u : 0u;     // what the NEXT fold does with this sequence's counters
    f->p->prev_bounces = f->max_bounces;
    f->p->cur_slots = n_slots;
    HIPCHK(ctx, hipGetLastError());
    return RT_OK;
}
} // namespace

extern "C" {

static int fold_adapt_hook(rt_frame* f);

} // extern "C"

namespace
{
// ---- RT_OPT_FRAME_KERNEL: the stage API's sample as one launch (k_frame) ---------------------------------------------------------
// Can this frame's next sample go through k_frame?  One sample in flight over the whole tile in one chunk on one pipe, the full log layout,
// the wide tree in place, no per-frame feature that reads between the stages, and none of the paths k_frame has no instance for.
bool frame_kernel_eligible(const rt_frame* f)
{
    const rt_ctx* ctx = f->ctx;
    const uint32_t n_local = f->n_local ? f->n_local : 1u;
    // (fk_auto.frames has been advanced past the frame being started: frame k = frames - 1)
    const int fk = f->fk_auto.frames - 1;
    const bool wanted = f->frame_kernel == 255u ? (f->fk_auto.decided ? f->fk_auto.use_kernel : (!f->fk_auto.skip && (fk == 2 || fk == 3 || (fk >= 4 && (fk & 1))))) : f->frame_kernel != 0u;
    return wanted && f->n_local != 0u && !(f->denoiser || f->aov != 0) && ctx->scene.wide_ok && !ctx->scene.slow_shadow &&
           ctx->scene.d.emissive_nee == 0u && f->log_ovf_blocks == 0u && f->n_pipes == 1u && f->chunk_pixels >= n_local && f->stage_chunks <= 1u &&
           (f->trace_variant == 5u || f->trace_variant == 10u) && !f->profile && !f->timeline && f->select_form_box == 0u && f->max_bounces < 63u;
}

template <bool FURNACE, bool BLUE>
int launch_frame_kernel_t(rt_frame* f)
{
    rt_ctx* ctx = f->ctx;
    PathPipe& q = f->ps[0];
    if (f->frame_blocks == 0u)
    {
        // RT_OPT_FRAME_KERNEL = 1: as many one-wave blocks as the device keeps resident (the kernel's registers decide), every wave with the same
        // number of chunks; = k >= 2: k chunks per wave, i.e. MORE blocks than are resident -- the hardware starts the next block where one has
        // finished, which is a dynamic schedule of the frame's chunks in units of k
        int per_cu = 0;
        HIPCHK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_frame<FURNACE, BLUE>, 64, 0));
        if (per_cu < 1) return fail(ctx, "rt_advance_sample: k_frame does not fit the device");
        const uint32_t n_chunks = (f->n_local + 63u) >> 6;
        const uint32_t resident = std::max(8u, (uint32_t)ctx->prop.multiProcessorCount * (uint32_t)per_cu);
        const uint32_t cpw = f->frame_kernel >= 2u && f->frame_kernel != 255u ? f->frame_kernel : (n_chunks + resident - 1u) / resident;     // chunks per wave
        //  all three or none: the grid's size is committed only with its buffers, so a failed allocation cannot leave a later
        // launch with frame_blocks != 0 and a NULL buffer
        const uint32_t n_blocks = (((n_chunks + cpw - 1u) / cpw) + 7u) & ~7u;
        uint32_t *counts = nullptr, *slow = nullptr; uint2* spill = nullptr;
        const bool ok = f->debug_alloc_limit != 0xFFFFFFFFu &&                       // (RT_OPT_DEBUG_ALLOC_LIMIT = 0xFFFFFFFF: the test hook for this path)
                        hipMalloc((void**)&counts, (size_t)n_blocks * RT_FRAME_COUNT_STRIDE * sizeof(uint32_t)) == hipSuccess &&
                        hipMalloc((void**)&slow, (size_t)n_blocks * cpw * 64u * sizeof(uint32_t)) == hipSuccess &&
                        hipMalloc((void**)&spill, (size_t)n_blocks * 64u * (RT_W4_STACK_MAX - 12) * sizeof(uint2)) == hipSuccess;
        if (!ok)
        {
            (void)hipGetLastError();
            for (void* p : {(void*)counts, (void*)slow, (void*)spill}) if (p) (void)hipFree(p);
            return fail(ctx, "rt_advance_sample: out of device memory for k_frame's per-wave buffers");
        }
        f->frame_counts = counts; f->frame_slow = slow; f->frame_spill = spill;
        f->frame_blocks = n_blocks;
        f->frame_chunks_per_wave = cpw;
    }
    // the previous sample's per-bounce counters go to the totals first (k_raygen does this for the stage kernels)
    hipLaunchKernelGGL(k_fold_counters, dim3(1), dim3(64), 0, q.stream, q.counters, q.prev_bounces, q.fold_accumulates);
    q.fold_accumulates = 0;
    q.prev_bounces = f->max_bounces;
    FrameArgs fa;
    memset(&fa, 0, sizeof(fa));
    ShadeArgs& a = fa.shade;
    a.log = dlog(f); a.counters = q.counters;
    a.bn_sobol = ctx->blue_noise; a.bn_scramble = ctx->blue_noise ? ctx->blue_noise + 65536 : nullptr;
    a.bn_rank = ctx->blue_noise ? ctx->blue_noise + 65536 + 131072 : nullptr;
    a.sample_base = f->sample_count;
    a.n_local = f->chunk_pixels ? f->chunk_pixels : 1;
    a.pix_base = 0;
    a.count_in_ray = 1u;
    a.partition = 0;
    for (int i = 0; i < 2; ++i) { fa.o4[i] = q.o4[i]; fa.d4[i] = q.d4[i]; fa.thr[i] = q.thr[i]; }
    fa.hits = q.hits;
    fa.sh_o4 = q.sh_o4[0]; fa.sh_d4 = q.sh_d4[0]; fa.sh_aux = q.sh_aux[0];
    fa.radiance = f->radiance;
    fa.spill = f->frame_spill;
    fa.slow_list = f->frame_slow;
    fa.wave_counts = f->frame_counts;
    fa.cam = f->camera;
    fa.tan_half_fov = rt_tanf(0.5f * f->camera.fov);
    fa.max_bounces = f->max_bounces;
    fa.drop_last = f->drop_last;
    const uint32_t t = f->trace_tune;
    fa.tune = ((t & 0xFFu) ? (t & 0xFFu) : (RT_TRACE2_DEFAULT_TUNE & 0xFFu)) | ((((t >> 8) & 0xFFu) ? ((t >> 8) & 0xFFu) : ((RT_TRACE2_DEFAULT_TUNE >> 8) & 0xFFu)) << 8);
    fa.tail_q = f->trace_tail_lanes;
    fa.chunks_per_wave = f->frame_chunks_per_wave;
    hipLaunchKernelGGL((k_frame<FURNACE, BLUE>), dim3(f->frame_blocks), dim3(64), 0, q.stream, ctx->scene.d, f->tile, fa);
    hipLaunchKernelGGL(k_frame_sum, dim3(130), dim3(256), 0, q.stream, (const uint32_t*)f->frame_counts, f->frame_blocks, f->max_bounces, q.counters);
    HIPCHK(ctx, hipGetLastError());
    q.chunk_base = 0;
    q.chunk_count = f->n_local;
    ++f->frame_launches;
    return RT_OK;
}

int launch_frame_kernel(rt_frame* f)
{
    const bool blue = f->sampler == 1;
    if (f->white_furnace) return blue ? launch_frame_kernel_t<true, true>(f) : launch_frame_kernel_t<true, false>(f);
    return blue ? launch_frame_kernel_t<false, true>(f) : launch_frame_kernel_t<false, false>(f);
}

} // namespace

extern "C" {

// The recorded stages of a deferred sample, run with the stage kernels after all (somebody wants the state between two stages).
static int deferred_materialize(rt_frame* f)
{
    if (!f->deferred.active) return RT_OK;
    const uint32_t done = f->deferred.bounce;
    const int next = f->deferred.next;
    f->deferred.active = false;
    // RT_OPT_SAMPLES_AHEAD: somebody looks between the stages of a sample that sits in a bank -- the frame traces it itself after all, and what
    // was traced ahead (that sample's slot first of all) is dropped
    if (f->deferred.ahead) { f->deferred.ahead = false; ahead_discard(f); }
    if (generate_rays(f, 1) != RT_OK) return RT_ERROR;
    for (uint32_t b = 0; b <= done; ++b)
    {
        const int upto = b < done ? 3 : next;                            // stages of bounce b that were recorded
        if (upto >= 1 && rt_intersect(f, b) != RT_OK) return RT_ERROR;
        if (upto >= 2 && rt_shade(f, b) != RT_OK) return RT_ERROR;
        if (upto >= 3 && rt_intersect_shadow(f, b) != RT_OK) return RT_ERROR;
    }
    return RT_OK;
}

int rt_generate_rays(rt_frame* f)                       // GenerateRays, :516-520
{
    FRAME_PROLOGUE(f, "rt_generate_rays");
    // the reference's own pattern (one Integrate() per frame through the hooks) adapts its folds too: probe, worker and adoption ride on
    // the frames' first stage (rt_integrate calls the same hook)
    if (fold_adapt_hook(f) != RT_OK) return RT_ERROR;
    // The stages run on the FULL log layout.  A compact log's pool can run dry (k_shade raises DCounters::log_ovf_flag); rt_integrate then repeats the batch in the
    // full layout, but here the caller drives the stages and there is nothing to repeat -- the sample's later entries would simply be missing from the sum (fuzz seed
    // 5652 of a 10 000-seed campaign, round 6: RT_OPT_COMPACT_LOG + a test-sized pool + the stage API, 20 pixels short).  One reallocation, the first time a frame
    // whose buffers are compact is driven through the stages.
    if (f->log_ovf_blocks != 0u)
    {
        if (f->p->cur_slots != 0 || f->deferred.active) return fail(ctx, "rt_generate_rays: the previous sample was not advanced (rt_advance_sample)");
        if (flush_log(f) != RT_OK || join_pipes(f) != RT_OK) return RT_ERROR;
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        f->log_full_forced = true;
        if (alloc_path_buffers(f, f->slots ? f->slots : 1u) != RT_OK) return RT_ERROR;
    }
    const uint32_t n_local = f->n_local ? f->n_local : 1u;
    uint32_t np = 1, cp = 0;
    chunk_plan(f, 1, np, cp);
    if ((uint64_t)cp * np < n_local)
        return fail(ctx, "rt_generate_rays: RT_OPT_PATH_STATE_LIMIT_MB is too small for one sample of the whole tile "
                         "(the stage API keeps every chunk on a pipe of its own; use rt_integrate)");
    if (f->deferred.active) return fail(ctx, "rt_generate_rays: the previous sample was not advanced (rt_advance_sample)");
    if (np <= 1)
    {
        f->stage_chunks = 1;
        if (ensure_whole_tile(f) != RT_OK) return RT_ERROR;
        if (f->ahead && ahead_wanted(f) && ahead_holds(f, f->sample_count) >= 0)
        {
            // RT_OPT_SAMPLES_AHEAD: this sample has been traced ahead -- nothing is launched, the stages are recorded, rt_advance_sample replays its log
            if (f->p->cur_slots != 0) return fail(ctx, "rt_generate_rays: the previous sample was not advanced (rt_advance_sample)");
            f->deferred.active = true; f->deferred.bounce = 0; f->deferred.next = 0; f->deferred.ahead = true;
            return RT_OK;
        }
        if (f->frame_kernel == 255u)
        {
            // frames 0 - 3 warm up (two each way), frames 4 - 19 alternate stage kernels / k_frame and are timed; once the last timed frame's events
            // have completed, the faster way stays (until the next scene upload)
            auto& m = f->fk_auto;
            if (m.scene != ctx->scene_uploads) { m.frames = 0; m.timing = -1; m.decided = false; m.use_kernel = false; m.scene = ctx->scene_uploads; }
            m.timing = -1;
            m.skip = false;
            if (!m.decided)
            {
                const int k = m.frames;
                if (k >= 20 && hipEventQuery(m.ev[15][1]) == hipSuccess)
                {
                    m.ms_stage = m.ms_kernel = 0.0f;
                    bool ok = true;
                    for (int i = 0; i < 16 && ok; ++i)
                    {
                        float t = 0.0f;
                        ok = hipEventElapsedTime(&t, m.ev[i][0], m.ev[i][1]) == hipSuccess;
                        ((i & 1) ? m.ms_kernel : m.ms_stage) += t;
                    }
                    if (!ok) (void)hipGetLastError();
                    m.decided = true;
                    m.use_kernel = ok && m.ms_kernel < m.ms_stage;
                }
                else if (k >= 4 && k < 20 && ctx->scene.adapt && (ctx->scene.adapt->state == FoldAdapt::ARMED || ctx->scene.adapt->state == FoldAdapt::PROBING || ctx->scene.adapt->state == FoldAdapt::COMPUTING))
                {
                    //  a fold adaptation is under way -- probe launches on this stream, a fold swapped mid-window: such a frame is not one
                    // of the sixteen that are compared (it runs through the stage kernels, untimed; the schedule goes on once the folds have settled)
                    m.skip = true;
                }
                else if (k >= 4 && k < 20)
                {
                    m.timing = k - 4;
                    bool ok = true;
                    for (hipEvent_t& e : m.ev[m.timing]) if (!e) ok = ok && hipEventCreate(&e) == hipSuccess;
                    if (!ok || hipEventRecord(m.ev[m.timing][0], ctx->stream) != hipSuccess) { (void)hipGetLastError(); m.decided = true; m.use_kernel = false; m.timing = -1; }
                }
                if (!m.decided && !m.skip) ++m.frames;
            }
            if (m.decided) m.frames = 1 << 20;          // (frame_kernel_eligible reads `decided`; keep frames - 1 out of the schedule's range)
        }
        if (frame_kernel_eligible(f))
        {
            // RT_OPT_FRAME_KERNEL: nothing is launched yet -- the stages are recorded, rt_advance_sample launches k_frame
            if (f->p->cur_slots != 0) return fail(ctx, "rt_generate_rays: the previous sample was not advanced (rt_advance_sample)");
            if (ensure_slots(f, 1) != RT_OK) return RT_ERROR;
            if (frame_kernel_eligible(f)) { f->deferred.active = true; f->deferred.bounce = 0; f->deferred.next = 0; f->deferred.ahead = false; return RT_OK; }
        }
        return generate_rays(f, 1);
    }
    // RT_OPT_STAGE_PIPES: the sample's chunks travel side by side, one per pipe; an allocation made for a larger batch (one pipe, the
    // whole tile) gives way to the one-sample layout
    if (f->chunk_pixels != cp || f->n_pipes != np)
    {
        for (const PathPipe& q : f->ps) if (q.cur_slots != 0) return fail(ctx, "rt_generate_rays: the previous sample was not advanced (rt_advance_sample)");
        if (flush_log(f) != RT_OK || join_pipes(f) != RT_OK) return RT_ERROR;
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (alloc_path_buffers(f, 1) != RT_OK) return RT_ERROR;
        if (f->chunk_pixels != cp || f->n_pipes != np) return fail(ctx, "rt_generate_rays: the one-sample layout did not come out as planned");
    }
    if (fork_pipes(f) != RT_OK) return RT_ERROR;         // the pipes' streams see what the context's stream has been given (reset, the last resolve)
    const uint32_t n_chunks = (n_local + f->chunk_pixels - 1u) / f->chunk_pixels;
    f->stage_chunks = n_chunks;
    uint32_t c = 0;
    return for_stage_pipes(f, [&]() { return generate_rays(f, 1, (c++) * f->chunk_pixels, false); });
}

int rt_intersect(rt_frame* f, uint32_t bounce)          // IntersectRays, :522-539
{
    FRAME_PROLOGUE(f, "rt_intersect");
    if (bounce > RT_MAX_BOUNCES_LIMIT) return fail(ctx, "rt_intersect: bounce out of range");
    if (f->deferred.active)
    {
        if (f->deferred.next == 0 && bounce == f->deferred.bounce) { f->deferred.next = 1; return RT_OK; }
        if (deferred_materialize(f) != RT_OK) return RT_ERROR;
    }
    uint32_t in = bounce & 1u;
    f->timeline_bounce = bounce;
    auto one = [&]() -> int
    {
        f->tl_stream = f->p->stream; f->tl_spill = f->p->spill; f->tl_slow_list = f->p->slow_list; f->tl_flavour = 0;
        KernelSpan span(f, 1);
        launch_trace<false>(f, f->p->o4[in], f->p->d4[in], (const uint32_t*)nullptr, &f->p->counters->queue[bounce], bounce);
        HIPCHK(ctx, hipGetLastError());
        return RT_OK;
    };
    return f->fused ? one() : for_stage_pipes(f, one);                   // (rt_integrate walks its pipes itself)
}

int rt_shade_miss(rt_frame* f, uint32_t) { return f ? RT_OK : fail(nullptr, "rt_shade_miss: frame is NULL"); }
int rt_clear_outgoing_counter(rt_frame* f, uint32_t) { return f ? RT_OK : fail(nullptr, "rt_clear_outgoing_counter: frame is NULL"); }
int rt_clear_shadow_counter(rt_frame* f) { return f ? RT_OK : fail(nullptr, "rt_clear_shadow_counter: frame is NULL"); }
int rt_accumulate_direct(rt_frame* f) { return f ? RT_OK : fail(nullptr, "rt_accumulate_direct: frame is NULL"); }

int rt_shade(rt_frame* f, uint32_t bounce)              // ShadeMissedRays + ShadeSurfaceHits, :582-643
{
    FRAME_PROLOGUE(f, "rt_shade");
    if (bounce > RT_MAX_BOUNCES_LIMIT) return fail(ctx, "rt_shade: bounce out of range");
    if (2u * (bounce + 1u) > f->log_entries) return fail(ctx, "rt_shade: bounce beyond the configured max_bounces");
    if (f->deferred.active)
    {
        if (f->deferred.next == 1 && bounce == f->deferred.bounce) { f->deferred.next = 2; return RT_OK; }
        if (deferred_materialize(f) != RT_OK) return RT_ERROR;
    }
    auto one = [&]() -> int
    {
    uint32_t in = bounce & 1u, out = (bounce + 1u) & 1u;
    ShadeArgs a;
    a.in_o4 = f->p->o4[in]; a.in_d4 = f->p->d4[in]; a.in_thr = f->p->thr[in]; a.hits = f->p->hits;
    a.out_o4 = f->p->o4[out]; a.out_d4 = f->p->d4[out]; a.out_thr = f->p->thr[out];
    a.sh_o4 = f->p->sh_o4[bounce & 1u]; a.sh_d4 = f->p->sh_d4[bounce & 1u]; a.sh_aux = f->p->sh_aux[bounce & 1u];
    a.log = dlog(f); a.counters = f->p->counters;
    a.bn_sobol = ctx->blue_noise; a.bn_scramble = ctx->blue_noise ? ctx->blue_noise + 65536 : nullptr;
    a.bn_rank = ctx->blue_noise ? ctx->blue_noise + 65536 + 131072 : nullptr;
    a.bounce = bounce; a.sample_base = f->sample_count;
    a.emit_outgoing = (f->drop_last && bounce >= f->max_bounces) ? 0u : 1u;
    a.n_local = f->chunk_pixels ? f->chunk_pixels : 1;
    a.pix_base = f->p->chunk_base;
    a.count_in_ray = f->fused ? 1u : 0u;
    a.partition = f->shade_partition;
    a.final_bounce = bounce >= f->max_bounces ? 1u : 0u;
    uint32_t blocks = (f->p->chunk_count * (f->p->cur_slots ? f->p->cur_slots : 1u) + RT_SHADE_BLOCK - 1u) / RT_SHADE_BLOCK;
    if (blocks == 0) blocks = 1;
    f->p->shadow_pending = true;
    // this bounce refills shadow queue [bounce & 1] and rewinds its work heads: the shadow trace of bounce - 2 is done with them
    if (wait_shadow(f, bounce & 1u) != RT_OK) return RT_ERROR;
    KernelSpan span(f, 2);
    const bool blue = f->sampler == 1;   // kernel variants are AOT (the reference rebuilds with -D..., :267-285)
    const bool nee = ctx->scene.d.emissive_nee != 0;     // opt-in extension (rt_scene_desc::flags)
#define RT_LAUNCH_SHADE(FURNACE, BLUE, NEE) \
    do { if (f->log_ovf_blocks) hipLaunchKernelGGL((k_shade<FURNACE, BLUE, NEE, true>), dim3(blocks), dim3(RT_SHADE_BLOCK), 0, f->p->stream, ctx->scene.d, f->tile, a); \
         else hipLaunchKernelGGL((k_shade<FURNACE, BLUE, NEE, false>), dim3(blocks), dim3(RT_SHADE_BLOCK), 0, f->p->stream, ctx->scene.d, f->tile, a); } while (0)
    if (nee)
    {
        if (f->white_furnace && blue) RT_LAUNCH_SHADE(true, true, true);
        else if (f->white_furnace) RT_LAUNCH_SHADE(true, false, true);
        else if (blue) RT_LAUNCH_SHADE(false, true, true);
        else RT_LAUNCH_SHADE(false, false, true);
    }
    else if (f->white_furnace && blue) RT_LAUNCH_SHADE(true, true, false);
    else if (f->white_furnace) RT_LAUNCH_SHADE(true, false, false);
    else if (blue) RT_LAUNCH_SHADE(false, true, false);
    else RT_LAUNCH_SHADE(false, false, false);
#undef RT_LAUNCH_SHADE
    HIPCHK(ctx, hipGetLastError());
    f->side_active = side_on(f);
    if (f->side_active) HIPCHK(ctx, hipEventRecord(f->p->ev_shaded, f->p->stream));
    return RT_OK;
    };
    return f->fused ? one() : for_stage_pipes(f, one);
}

int rt_intersect_shadow(rt_frame* f, uint32_t bounce)   // IntersectShadowRays + AccumulateDirectSamples, :564-580,645-649
{
    FRAME_PROLOGUE(f, "rt_intersect_shadow");
    if (bounce > RT_MAX_BOUNCES_LIMIT) return fail(ctx, "rt_intersect_shadow: bounce out of range");
    if (f->deferred.active)
    {
        if (f->deferred.next == 2 && bounce == f->deferred.bounce) { f->deferred.next = 0; ++f->deferred.bounce; return RT_OK; }
        if (deferred_materialize(f) != RT_OK) return RT_ERROR;
    }
    const uint32_t q = bounce & 1u;
    auto one = [&]() -> int
    {
        f->tl_stream = f->side_active ? f->p->side : f->p->stream;
        f->tl_spill = f->p->sh_spill; f->tl_slow_list = f->p->sh_slow_list; f->tl_flavour = 1u + q;
        if (f->side_active) HIPCHK(ctx, hipStreamWaitEvent(f->p->side, f->p->ev_shaded, 0));
        {
            KernelSpan span(f, 3, f->tl_stream);
            launch_trace<true>(f, f->p->sh_o4[q], f->p->sh_d4[q], (const uint32_t*)f->p->sh_aux[q], &f->p->counters->shadow[bounce], bounce);
        }
        f->p->shadow_pending = false;
        HIPCHK(ctx, hipGetLastError());
        if (f->side_active)
        {
            HIPCHK(ctx, hipEventRecord(f->p->ev_shadow[q], f->p->side));
            f->p->shadow_in_flight[q] = true;
        }
        return RT_OK;
    };
    return f->fused ? one() : for_stage_pipes(f, one);
}

int rt_compute_aovs(rt_frame* f)                        // ComputeAOVs, :541-562 (after rt_intersect(frame, 0))
{
    FRAME_PROLOGUE(f, "rt_compute_aovs");
    if (f->aov == 0 && !f->denoiser) return RT_OK;      // outputs unobservable: skip the work
    if (f->p->cur_slots != 1) return fail(ctx, "rt_compute_aovs: AOVs need one sample in flight (use the stage API or rt_integrate with the denoiser/AOV option set)");
    if (f->n_local == 0) return RT_OK;
    uint32_t blocks = (f->n_local + 255u) / 256u;
    hipLaunchKernelGGL(k_aov_clear, dim3(blocks), dim3(256), 0, ctx->stream, f->aov_buf, f->n_local);
    hipLaunchKernelGGL(k_aov, dim3(blocks), dim3(256), 0, ctx->stream, ctx->scene.d, (const float4*)f->p->o4[0],
        (const float4*)f->p->d4[0], (const float4*)f->p->hits, (const uint32_t*)&f->p->counters->queue[0], f->camera, f->prev_camera,
        rt_tanf(0.5f * f->camera.fov), rt_tanf(0.5f * f->prev_camera.fov), f->aov_buf);
    HIPCHK(ctx, hipGetLastError());
    return RT_OK;
}

int rt_denoise(rt_frame* f)                             // Denoise, :665-668
{
    FRAME_PROLOGUE(f, "rt_denoise");
    if (f->denoiser != 1 || f->n_local == 0) return RT_OK;   // 2: rt_group_denoise does it on the gathered image
    if (flush_log(f) != RT_OK) return RT_ERROR;
    uint32_t blocks = (f->n_local + 255u) / 256u;
    hipLaunchKernelGGL(k_denoise, dim3(blocks), dim3(256), 0, ctx->stream, f->tile.width, f->tile.height, f->radiance,
        (const float4*)f->prev_radiance, (const float*)f->aov_buf.depth, (const float*)f->prev_depth,
        (const float2*)f->aov_buf.velocity);
    HIPCHK(ctx, hipGetLastError());
    return RT_OK;
}

int rt_copy_history(rt_frame* f)                        // CopyHistoryBuffers, :670-675
{
    FRAME_PROLOGUE(f, "rt_copy_history");
    if (f->denoiser != 1 || f->n_local == 0) return RT_OK;
    HIPCHK(ctx, hipMemcpyAsync(f->prev_radiance, f->radiance, (size_t)f->n_local * sizeof(float4), hipMemcpyDeviceToDevice,
        ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(f->prev_depth, f->aov_buf.depth, (size_t)f->n_local * sizeof(float), hipMemcpyDeviceToDevice,
        ctx->stream));
    return RT_OK;
}

#include "samples_ahead_impl.h"

int rt_advance_sample(rt_frame* f)                      // AdvanceSampleCount, :510-514
{
    if (!f) return fail(nullptr, "rt_advance_sample: frame is NULL");
    (void)hipSetDevice(f->ctx->device);
    if (f->deferred.active)
    {
        const bool whole = f->deferred.next == 0 && f->deferred.bounce == f->max_bounces + 1u;
        if (f->deferred.ahead)
        {
            // RT_OPT_SAMPLES_AHEAD: the sample was traced ahead; its recorded stages are dropped and its log slot is replayed
            const int bank = whole ? ahead_holds(f, f->sample_count) : -1;
            if (bank >= 0)
            {
                f->deferred.active = false; f->deferred.ahead = false;
                if (ahead_consume(f, bank) != RT_OK) return RT_ERROR;
                ++f->ahead->quiet;
                ahead_schedule(f, false);
                return RT_OK;
            }
        }
        // the whole sample was recorded in the canonical order: ONE launch (k_frame replays its own pixels' log, too)
        else if (whole && frame_kernel_eligible(f))
        {
            if (launch_frame_kernel(f) != RT_OK)
            {
                //  the recorded sample is not lost: it runs through the stage kernels, and this frame stops asking for k_frame
                (void)hipGetLastError();
                f->frame_kernel = 0u;
                f->fk_auto.decided = true; f->fk_auto.use_kernel = false; f->fk_auto.timing = -1;
                if (deferred_materialize(f) != RT_OK || flush_stage(f) != RT_OK) return RT_ERROR;
                f->sample_count += 1;
                ahead_after_plain_sample(f);
                return RT_OK;
            }
            f->deferred.active = false;
            f->sample_count += 1;
            if (f->fk_auto.timing >= 0) { (void)hipEventRecord(f->fk_auto.ev[f->fk_auto.timing][1], f->ctx->stream); f->fk_auto.timing = -1; }
            ahead_after_plain_sample(f);
            return RT_OK;
        }
        if (deferred_materialize(f) != RT_OK) return RT_ERROR;
    }
    uint32_t n = f->p->cur_slots ? f->p->cur_slots : 1u;
    if (flush_stage(f) != RT_OK) return RT_ERROR;        // radiance_buffer_ += this sample's contributions (of every chunk: RT_OPT_STAGE_PIPES)
    f->sample_count += n;
    if (f->fk_auto.timing >= 0) { (void)hipEventRecord(f->fk_auto.ev[f->fk_auto.timing][1], f->ctx->stream); f->fk_auto.timing = -1; }   // RT_OPT_FRAME_KERNEL = 255
    ahead_after_plain_sample(f);
    return RT_OK;
}

int rt_frame_reserve_samples(rt_frame* f, uint32_t n_samples, uint32_t* reserved)
{
    FRAME_PROLOGUE(f, "rt_frame_reserve_samples");
    f->reserved_explicitly = true;                       // (rt_integrate then takes the buffers as they are: no lean growth)
    if (ensure_slots(f, n_samples ? n_samples : 1u) != RT_OK) return RT_ERROR;
    if (reserved) *reserved = f->slots;
    return RT_OK;
}

#include "fold_hooks_impl.h"

int rt_integrate(rt_frame* f, uint32_t n_samples)       // n x Integrator::Integrate(), integrator.cpp:27-59
{
    FRAME_PROLOGUE(f, "rt_integrate");
    if (f->deferred.active) return fail(ctx, "rt_integrate: a sample of the stage API is in flight (rt_advance_sample first)");
    if (f->stage_chunks > 1)
    {
        for (const PathPipe& q : f->ps) if (q.cur_slots != 0) return fail(ctx, "rt_integrate: a sample of the stage API is in flight (rt_advance_sample first)");
        f->stage_chunks = 1;
    }
    ahead_discard(f);                                   // RT_OPT_SAMPLES_AHEAD serves the stage API; these samples are traced here
    if (fold_adapt_hook(f) != RT_OK) return RT_ERROR;
    // `slots` samples travel through the wavefront together (more rays per launch ->
    // fuller machine, shorter relative tails); the radiance log keeps the sum exact.
    uint32_t done = 0;
    const bool per_frame = f->denoiser || f->aov != 0;  // interactive features: one sample per Integrate()
    uint32_t cap = per_frame ? 1u : slot_cap(f);
    uint32_t want = n_samples < cap ? n_samples : cap;
    // Lean growth (round 6).  Mapping the per-path buffers is not free: a hipMalloc of the 102 GB that 128 samples of a 1080p frame in flight take costs 2.4 - 3.5 s on
    // an MI355X box (bench.py: config.path_state_alloc_s, cold_job.alloc_s) -- six times what BASELINE's whole 256-spp job renders in.  So a job that has to GROW the buffers,
    // was not told how many samples to keep in flight (RT_OPT_SAMPLES_IN_FLIGHT = 0) and did not reserve them (rt_frame_reserve_samples) grows them with what it has been asked
    // for so far: an eighth of the samples requested since the frame was made (at least 16), never below what 8 GiB hold -- 256 spp at 1080p: 32 in flight, 27 GB, ~5 % below the
    // full batch's rate; a job that keeps asking reaches the full batch by doubling.
    f->samples_asked += n_samples;
    if (!per_frame && f->slots_opt == 0u && !f->reserved_explicitly && want > f->slots)
    {
        const uint64_t per_slot = (uint64_t)bytes_per_path(f, want) * (f->n_local ? f->n_local : 1u);
        const uint64_t lean = std::max<uint64_t>(16u, (f->samples_asked + 7u) / 8u), floor_slots = (8ull << 30) / (per_slot ? per_slot : 1u);
        const uint64_t grown = std::max<uint64_t>(std::max(lean, floor_slots), f->slots);
        if (grown < want) want = (uint32_t)grown;
    }
    if (ensure_slots(f, want) != RT_OK) return RT_ERROR;
    // ensure_slots may have halved the batch to fit the device (slots_limit): never ask for more than it got
    if (!per_frame) cap = slot_cap(f) < f->slots ? slot_cap(f) : f->slots;
    if (cap == 0) cap = 1;
    if (per_frame && chunk_for(f, 1) >= (f->n_local ? f->n_local : 1u) && ensure_whole_tile(f) != RT_OK) return RT_ERROR;
    if (per_frame && f->chunk_pixels < (f->n_local ? f->n_local : 1u))
        return fail(ctx, "rt_integrate: AOVs / the denoiser need the whole tile in one chunk (raise RT_OPT_PATH_STATE_LIMIT_MB)");
    if (fork_pipes(f) != RT_OK) return RT_ERROR;
    int rc = RT_OK;
    f->fused = true;
    f->side_active = side_on(f);
    while (done < n_samples && rc == RT_OK)
    {
        // batches of (nearly) equal size: 1024 samples with room for 160 in flight go as 7 x 146-147, not 6 x 160 + 64 --
        // every launch costs its tail whatever its size (DESIGN.md "Where a launch's time goes")
        const uint32_t left = n_samples - done, n_batches = (left + cap - 1u) / cap;
        uint32_t batch = (left + n_batches - 1u) / n_batches;
        if (per_frame) batch = 1;
        if (f->denoiser && rt_reset(f) != RT_OK) { rc = RT_ERROR; break; }   // integrator.cpp:29: Reset() every frame
        // The whole wavefront loop per chunk of pixels; chunk c runs on pipe c % n_pipes (its own stream), so the
        // chunks' launches overlap each other's tails.  A chunk always lands on the same pipe: its log replays
        // (k_flush) stay in sample order.
        uint32_t c = 0;
        for (uint32_t base = 0; base < (f->n_local ? f->n_local : 1u) && rc == RT_OK; ++c)
        {
            f->p = &f->ps[c % f->n_pipes];
            if (generate_rays(f, batch, base, c >= f->n_pipes) != RT_OK) { rc = RT_ERROR; break; }
            // Per bounce: closest-hit trace, k_shade, shadow trace.  The shadow trace of bounce b and the closest-hit
            // trace of bounce b + 1 both depend on k_shade(b) only, and k_shade(b + 1) on the latter only (the shadow
            // queue is double-buffered): with overlap_shadow the shadow trace goes to the pipe's side stream, AFTER the
            // next closest-hit launch has been enqueued -- it moves in as that launch's last rays drain and leaves as
            // k_shade(b + 1) moves in, so neither tail idles the machine (tools/launch_timeline.py: 0.75-0.95 ms each).
            if (rt_intersect(f, 0) != RT_OK) rc = RT_ERROR;
            for (uint32_t bounce = 0; bounce <= f->max_bounces && rc == RT_OK; ++bounce)
            {
                if (bounce == 0 && per_frame && rt_compute_aovs(f) != RT_OK) rc = RT_ERROR;
                else if (rt_shade(f, bounce) != RT_OK) rc = RT_ERROR;
                else if (f->side_active && bounce < f->max_bounces && rt_intersect(f, bounce + 1u) != RT_OK) rc = RT_ERROR;
                else if (rt_intersect_shadow(f, bounce) != RT_OK) rc = RT_ERROR;
                else if (!f->side_active && bounce < f->max_bounces && rt_intersect(f, bounce + 1u) != RT_OK) rc = RT_ERROR;
            }
            // Compact log layout: did this sequence run the overflow pool dry?  (The one host synchronisation of a batch; the
            // full layout has none.)  If so nothing of it has reached the radiance yet: drop it, switch the frame to the full
            // layout within the same memory, and run the SAME chunk again -- the sum stays exact.
            if (rc == RT_OK && f->log_ovf_blocks != 0u)
            {
                uint32_t dry = 0;
                if (wait_shadow(f, 0) != RT_OK || wait_shadow(f, 1) != RT_OK) { rc = RT_ERROR; break; }
                if (hipMemcpyAsync(&dry, &f->p->counters->log_ovf_flag, sizeof(dry), hipMemcpyDeviceToHost, f->p->stream) != hipSuccess ||
                    hipStreamSynchronize(f->p->stream) != hipSuccess)
                {
                    rc = fail(ctx, "rt_integrate: reading the log-pool flag failed");
                    break;
                }
                if (dry)
                {
                    const uint64_t held = (uint64_t)f->log_stride * bytes_per_path(f, f->slots) * f->n_pipes;
                    if (hipMemsetAsync(f->p->counters->queue, 0, sizeof(f->p->counters->queue) + sizeof(f->p->counters->shadow), f->p->stream) != hipSuccess ||
                        hipMemsetAsync(&f->p->counters->log_ovf_flag, 0, sizeof(uint32_t), f->p->stream) != hipSuccess ||
                        hipStreamSynchronize(f->p->stream) != hipSuccess)
                    {
                        rc = fail(ctx, "rt_integrate: discarding a batch failed");
                        break;
                    }
                    f->p->cur_slots = 0;
                    f->p->shadow_pending = false;
                    f->log_full_forced = true;
                    // the full layout takes more bytes per path: it may use what the library would have given it in the first
                    // place (half of the HBM), and beyond that runs the same batch size in more chunks
                    f->fallback_limit_mb = 144u << 10;
                    (void)held;
                    ++f->log_fallbacks;
                    if (alloc_path_buffers(f, f->slots) != RT_OK) { rc = RT_ERROR; break; }     // cnt starts from zero again
                    continue;                                                                   // the same `base`, the new chunk size
                }
            }
            if (rc == RT_OK && flush_log(f) != RT_OK) rc = RT_ERROR;          // radiance_buffer_ += this chunk's contributions
            base += f->chunk_pixels;
        }
        f->p = &f->ps[0];
        if (rc != RT_OK) break;
        f->sample_count += batch;                                            // AdvanceSampleCount, :510-514
        if (f->denoiser && (join_pipes(f) != RT_OK || rt_denoise(f) != RT_OK || rt_copy_history(f) != RT_OK)) rc = RT_ERROR;
        done += batch;
    }
    f->p = &f->ps[0];
    f->fused = false;
    if (rc != RT_OK)
        for (PathPipe& q : f->ps)                        // a failed batch leaves no shadow trace running behind the caller's back
        {
            if (q.side) (void)hipStreamSynchronize(q.side);
            q.shadow_in_flight[0] = q.shadow_in_flight[1] = false;
        }
    if (join_pipes(f) != RT_OK) return RT_ERROR;         // whatever follows on the context's stream sees every chunk
    return rc;
}

// ---- output ----------------------------------------------------------------
int rt_frame_present_wait(rt_frame* f)
{
    if (!f) return fail(nullptr, "rt_frame_present_wait: frame is NULL");
    if (!f->present_pending) return RT_OK;
    (void)hipSetDevice(f->ctx->device);
    HIPCHK(f->ctx, hipEventSynchronize(f->ev_copied[(f->present_flip ^ 1u) & 1u]));      // the latest copy (its stream carries later frames' work too)
    f->present_pending = false;
    return RT_OK;
}

// ResolveRadiance + Finish() AS THE REFERENCE HAS THEM (cl_pt_integrator.cpp:677-684): the kernels have finished when this
// returns; the resolved image is on its way.  The reference resolves into a GL image its window blits later -- nothing crosses
// PCIe at all; headless, the image has to reach the host, and it does so on a copy stream of its own, double-buffered on the
// device, while the next frame's rays are already being traced (0.6 of the 4.7 ms of a 1080p frame were this read-back).
int rt_frame_present(rt_frame* f, float* host_rgba)
{
    if (!f || !host_rgba) return fail(nullptr, "rt_frame_present: NULL argument");
    rt_ctx* ctx = f->ctx;
    (void)hipSetDevice(ctx->device);
    if (f->n_local == 0) return RT_OK;
    if (!f->present_stream)
    {
        // all or nothing: a frame whose second image could not be allocated keeps presenting through rt_frame_resolve's path
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        hipStream_t st = nullptr;
        hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
        float4* second = nullptr;
        bool ok = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, lo) == hipSuccess;           // (used only by a frame without a side stream)
        for (hipEvent_t& e : ev) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipMalloc((void**)&second, (size_t)f->n_local * sizeof(float4)) == hipSuccess;
        ok = ok && hipEventRecord(ev[2], st) == hipSuccess && hipEventRecord(ev[3], st) == hipSuccess;
        if (!ok)
        {
            (void)hipGetLastError();
            if (second) (void)hipFree(second);
            for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
            if (st) (void)hipStreamDestroy(st);
            return rt_frame_resolve(f, host_rgba);       // the synchronous form: same image, no overlap
        }
        f->present_stream = st; f->resolved_b = second;
        f->ev_resolved[0] = ev[0]; f->ev_resolved[1] = ev[1]; f->ev_copied[0] = ev[2]; f->ev_copied[1] = ev[3];
    }
    if (flush_stage(f) != RT_OK) return RT_ERROR;
    const uint32_t i = f->present_flip & 1u;
    float4* image = i ? f->resolved_b : f->resolved;
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, f->ev_copied[i], 0));      // the copy that last read this device image (two frames ago)
    uint32_t blocks = (f->n_local + 255u) / 256u;
    hipLaunchKernelGGL(k_resolve, dim3(blocks), dim3(256), 0, ctx->stream, (const float4*)f->radiance, f->aov_buf,
        image, f->n_local, f->sample_count, f->aov, f->denoiser);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipEventRecord(f->ev_resolved[i], ctx->stream));
    // The copy goes to pipe 0's SIDE stream (where the shadow traces run), not to a stream of its own: which hardware queue a further stream lands
    // on is the runtime's business, and when it was the side stream's the copy's completion barrier held the next frame's shadow traces back for
    // the whole 0.6 ms -- in rt_render's process with a low-priority copy stream (3.87 instead of 3.33 ms per frame), in bench.py's (PyTorch has
    // made streams before) with a normal-priority one (3.86 instead of 3.38): profiles/r05_call19.log.  On the side stream itself the order is
    // explicit and harmless: the copy is enqueued at the end of frame N and runs while frame N + 1 generates, traces and shades its camera rays --
    // longer than the copy takes -- before that frame's first shadow trace is enqueued behind it.  3.30 / 3.31 ms per frame in both processes.
    hipStream_t const copy_stream = f->ps[0].side ? f->ps[0].side : f->present_stream;
    HIPCHK(ctx, hipStreamWaitEvent(copy_stream, f->ev_resolved[i], 0));
    // (The runtime's copy of a page-locked destination is a blit kernel, 441 us for a 1080p image; a 64-block copy kernel of our
    // own that left the other CUs alone was tried and is SLOWER end to end -- it holds PCIe for milliseconds and every persistent
    // grid launched meanwhile finds part of its residency taken: 4.22 instead of 3.70 ms per frame, profiles/r04_call07_*.  The
    // stream's low priority is what keeps the blit's workgroups behind the next frame's first launches.)
    HIPCHK(ctx, hipMemcpyAsync(host_rgba, image, (size_t)f->n_local * sizeof(float4), hipMemcpyDeviceToHost, copy_stream));
    HIPCHK(ctx, hipEventRecord(f->ev_copied[i], copy_stream));
    f->present_flip ^= 1u;
    f->present_pending = true;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));      // Finish(), :682: every kernel of the frame has run
    return RT_OK;
}

int rt_frame_resolve(rt_frame* f, float* host_rgba)     // ResolveRadiance, :677-684
{
    if (!f || !host_rgba) return fail(nullptr, "rt_frame_resolve: NULL argument");
    rt_ctx* ctx = f->ctx;
    (void)hipSetDevice(ctx->device);
    if (f->n_local == 0) return RT_OK;
    if (rt_frame_present_wait(f) != RT_OK) return RT_ERROR;                // an image still travelling to (possibly) the same host buffer
    if (flush_stage(f) != RT_OK) return RT_ERROR;
    uint32_t blocks = (f->n_local + 255u) / 256u;
    hipLaunchKernelGGL(k_resolve, dim3(blocks), dim3(256), 0, ctx->stream, (const float4*)f->radiance, f->aov_buf,
        f->resolved, f->n_local, f->sample_count, f->aov, f->denoiser);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(host_rgba, f->resolved, (size_t)f->n_local * sizeof(float4), hipMemcpyDeviceToHost,
        ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));      // the frame's only host sync, like Finish() at :682
    return RT_OK;
}

int rt_frame_read_radiance(rt_frame* f, float* host_rgba)
{
    if (!f || !host_rgba) return fail(nullptr, "rt_frame_read_radiance: NULL argument");
    rt_ctx* ctx = f->ctx;
    (void)hipSetDevice(ctx->device);
    if (f->n_local == 0) return RT_OK;
    if (flush_stage(f, true) != RT_OK) return RT_ERROR;
    HIPCHK(ctx, hipMemcpyAsync(host_rgba, f->radiance, (size_t)f->n_local * sizeof(float4), hipMemcpyDeviceToHost,
        ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

void* rt_frame_radiance_device_ptr(rt_frame* f) { return f ? (void*)f->radiance : nullptr; }
uint32_t rt_frame_sample_count(rt_frame* f) { return f ? f->sample_count : 0; }

int rt_frame_get_stats(rt_frame* f, rt_stats* out)
{
    if (!f || !out) return fail(nullptr, "rt_frame_get_stats: NULL argument");
    rt_ctx* ctx = f->ctx;
    (void)hipSetDevice(ctx->device);
    if (f->deferred.active && deferred_materialize(f) != RT_OK) return RT_ERROR;     // RT_OPT_FRAME_KERNEL: the counters of the stages recorded so far
    if (join_pipes(f) != RT_OK || sync_frame_streams(f) != RT_OK) return RT_ERROR;   // a shadow trace on a side stream still counts rays
    memset(out, 0, sizeof(*out));
    for (uint32_t i = 0; i < RT_MAX_PIPES; ++i)              // the pipes' counters add up
    {
        PathPipe& q = f->ps[i];
        if (!q.counters) continue;
        hipLaunchKernelGGL(k_fold_counters, dim3(1), dim3(64), 0, ctx->stream, q.counters, q.prev_bounces, q.fold_accumulates);
        q.fold_accumulates = 1;       // whatever is folded next belongs to the same batch's totals unless a new batch starts
        DCounters h;
        HIPCHK(ctx, hipMemcpyAsync(&h, q.counters, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        out->closest_rays += h.total_closest;
        out->shadow_rays += h.total_shadow;
        out->stack_spills += h.stack_spills;
        out->slow_rays += h.slow_rays;
        if (i < f->n_pipes)
            for (int b = 0; b < 64; ++b) { out->last_active[b] += h.last_queue[b]; out->last_shadow[b] += h.last_shadow[b]; }
    }
    out->samples = f->sample_count;
    out->samples_in_flight = f->slots;
    out->samples_in_flight_limit = f->slots_limit;
    out->path_state_bytes = (uint64_t)f->log_stride * bytes_per_path(f, f->log_ovf_blocks ? f->slots : 1u) * f->n_pipes;
    out->log_inline_entries = f->log_ovf_blocks ? f->log_inline : 0u;
    out->log_fallbacks = f->log_fallbacks;
    out->frame_kernel_samples = (uint32_t)f->frame_launches;
    out->chunk_pixels = f->chunk_pixels;
    out->pipelines = f->n_pipes;
    if (f->ahead)
    {
        // RT_OPT_SAMPLES_AHEAD: the rays of the samples that came out of the banks were counted there -- whole batches at a time, so the totals
        // include `samples_ahead` samples that `samples` does not yet
        for (const AheadBank& b : f->ahead->bank)
        {
            if (!b.h) continue;
            rt_stats hs;
            if (rt_frame_get_stats(b.h, &hs) != RT_OK) return RT_ERROR;
            out->closest_rays += hs.closest_rays; out->shadow_rays += hs.shadow_rays;
            out->stack_spills += hs.stack_spills; out->slow_rays += hs.slow_rays;
            out->path_state_bytes += hs.path_state_bytes;
            out->samples_ahead += b.n - b.next;
        }
        out->samples_from_banks = f->ahead->consumed;
    }
    return RT_OK;
}

int rt_frame_get_profile(rt_frame* f, rt_profile* out)
{
    if (!f || !out) return fail(nullptr, "rt_frame_get_profile: NULL argument");
    rt_ctx* ctx = f->ctx;
    (void)hipSetDevice(ctx->device);
    if (join_pipes(f) != RT_OK || sync_frame_streams(f) != RT_OK) return RT_ERROR;   // spans on the side streams end when their launches do
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    memset(out, 0, sizeof(*out));
    double* ms[4] = {&out->ms_raygen, &out->ms_trace_closest, &out->ms_shade, &out->ms_trace_shadow};
    uint32_t* cnt[4] = {&out->n_raygen, &out->n_trace_closest, &out->n_shade, &out->n_trace_shadow};
    for (auto& s : f->spans)
    {
        float t = 0.0f;
        if (hipEventElapsedTime(&t, s.a, s.b) == hipSuccess) { *ms[s.cls] += t; (*cnt[s.cls])++; }
        f->event_pool.push_back(s.a);
        f->event_pool.push_back(s.b);
    }
    f->spans.clear();
    return RT_OK;
}

int rt_frame_copy_radiance(rt_frame* f, void* device_dst)
{
    if (!f || !device_dst) return fail(nullptr, "rt_frame_copy_radiance: NULL argument");
    rt_ctx* ctx = f->ctx;
    (void)hipSetDevice(ctx->device);
    if (f->n_local == 0) return RT_OK;
    if (flush_stage(f, true) != RT_OK) return RT_ERROR;
    HIPCHK(ctx, hipMemcpyAsync(device_dst, f->radiance, (size_t)f->n_local * sizeof(float4), hipMemcpyDeviceToDevice,
        ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

// ---- debug / parity --------------------------------------------------------
int rt_frame_debug_read_queue(rt_frame* f, int which, uint32_t bounce, rt_ray* rays, uint32_t* pixel_indices,
    rt_float4* payload, uint32_t capacity, uint32_t* count)
{
    if (!f || !count) return fail(nullptr, "rt_frame_debug_read_queue: NULL argument");
    rt_ctx* ctx = f->ctx;
    (void)hipSetDevice(ctx->device);
    if (bounce > RT_MAX_BOUNCES_LIMIT + 1) return fail(ctx, "rt_frame_debug_read_queue: bounce out of range");
    if (f->stage_chunks > 1) return fail(ctx, "rt_frame_debug_read_queue: the sample in flight is spread over several pipes (set RT_OPT_STAGE_PIPES to 1 for the debug readers)");
    if (f->deferred.active && deferred_materialize(f) != RT_OK) return RT_ERROR;
    if (f->p->side) HIPCHK(ctx, hipStreamSynchronize(f->p->side));      // a shadow trace may be retracting log entries
    DCounters h;
    HIPCHK(ctx, hipMemcpyAsync(&h, f->p->counters, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    uint32_t n = which == 0 ? h.queue[bounce] : h.shadow[bounce];
    if (n > f->log_stride) return fail(ctx, "rt_frame_debug_read_queue: corrupt counter");
    *count = n;
    if (n == 0) return RT_OK;
    if (!rays && !pixel_indices && !payload) return RT_OK;                 // size query (two-call pattern)
    if (n > capacity) return fail(ctx, "rt_frame_debug_read_queue: the queue holds more entries than the caller's arrays");
    const float4* so = which == 0 ? f->p->o4[bounce & 1u] : f->p->sh_o4[bounce & 1u];
    const float4* sdir = which == 0 ? f->p->d4[bounce & 1u] : f->p->sh_d4[bounce & 1u];
    std::vector<float4> o(n), d(n), p(n);
    HIPCHK(ctx, hipMemcpy(o.data(), so, (size_t)n * 16, hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(d.data(), sdir, (size_t)n * 16, hipMemcpyDeviceToHost));
    if (which == 0) HIPCHK(ctx, hipMemcpy(p.data(), f->p->thr[bounce & 1u], (size_t)n * 16, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; ++i)
    {
        uint32_t pl;
        memcpy(&pl, &d[i].w, 4);
        uint32_t id = pl;
        uint32_t local_pix = f->p->chunk_base + id % (f->chunk_pixels ? f->chunk_pixels : 1);
        if (which == 1)   // the deferred direct-light sample lives in the radiance log
        {
            uint32_t entry = 0;
            HIPCHK(ctx, hipMemcpy(&entry, f->p->sh_aux[bounce & 1u] + i, 4, hipMemcpyDeviceToHost));
            p[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            uint32_t oblk = 0;
            if (entry >= f->log_inline) HIPCHK(ctx, hipMemcpy(&oblk, f->p->ovf_slot + id, 4, hipMemcpyDeviceToHost));
            const size_t at = entry < f->log_inline ? (size_t)entry * f->log_stride + id
                                                    : (size_t)f->log_inline * f->log_stride + (size_t)(entry - f->log_inline) * f->log_ovf_blocks + oblk;
            HIPCHK(ctx, hipMemcpy(&p[i], f->p->rlog + 3u * at, 12, hipMemcpyDeviceToHost));
        }
        if (rays)
        {
            rays[i].origin.x = o[i].x; rays[i].origin.y = o[i].y; rays[i].origin.z = o[i].z; rays[i].origin.w = 0.0f;
            rays[i].direction.x = d[i].x; rays[i].direction.y = d[i].y; rays[i].direction.z = d[i].z;
            rays[i].direction.w = o[i].w;
        }
        if (pixel_indices)
        {
            uint32_t ly = local_pix / f->tile.width, px = local_pix - ly * f->tile.width;
            pixel_indices[i] = rt_frame_global_row(f, ly) * f->tile.width + px;   // GLOBAL pixel index
        }
        if (payload) { payload[i].x = p[i].x; payload[i].y = p[i].y; payload[i].z = p[i].z; payload[i].w = 0.0f; }
    }
    return RT_OK;
}

int rt_frame_debug_read_hits(rt_frame* f, rt_hit* hits, uint32_t count)
{
    if (!f || !hits) return fail(nullptr, "rt_frame_debug_read_hits: NULL argument");
    rt_ctx* ctx = f->ctx;
    (void)hipSetDevice(ctx->device);
    if (count > f->log_stride) return fail(ctx, "rt_frame_debug_read_hits: count too large");
    if (f->stage_chunks > 1) return fail(ctx, "rt_frame_debug_read_hits: the sample in flight is spread over several pipes (set RT_OPT_STAGE_PIPES to 1 for the debug readers)");
    if (f->deferred.active && deferred_materialize(f) != RT_OK) return RT_ERROR;
    std::vector<float4> h(count ? count : 1);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpy(h.data(), f->p->hits, (size_t)count * 16, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < count; ++i)
    {
        hits[i].bc.x = h[i].x; hits[i].bc.y = h[i].y;
        memcpy(&hits[i].primitive_id, &h[i].z, 4);
        hits[i].t = h[i].w;
    }
    return RT_OK;
}

// Debug: launch timeline of the closest-hit wide-tree kernel.  arm = 1 clears the slots and starts recording
// (pipe 0); arm = 0 reads them: out[b] = {first wave started, first wave found the queue dry, last wave left} of the
// most recent bounce-b launch, in ticks of the 100 MHz wall clock (0 where nothing ran), then the most traversal steps
// any ray took and the slowest ray's ticks from hand-out to retirement and its steps; after those 64 x 6 values, out[384 + i] =
// waves (of all recorded launches) that left in the i-th 25 us after their launch's queue ran dry.
// k_frame's per-wave rows of its latest launch (RT_FRAME_COUNT_STRIDE words each; frame_kernels.h): rays per bounce and the wave's ticks per phase
int rt_frame_debug_frame_rows(rt_frame* f, uint32_t* out, uint32_t capacity_rows, uint32_t* n_rows, uint32_t* row_words)
{
    if (!f || !n_rows || !row_words) return fail(nullptr, "rt_frame_debug_frame_rows: NULL argument");
    rt_ctx* ctx = f->ctx;
    (void)hipSetDevice(ctx->device);
    *n_rows = f->frame_blocks; *row_words = RT_FRAME_COUNT_STRIDE;
    if (!out || f->frame_blocks == 0u || !f->frame_counts) return RT_OK;
    if (capacity_rows < f->frame_blocks) return fail(ctx, "rt_frame_debug_frame_rows: the caller's array is too small");
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpy(out, f->frame_counts, (size_t)f->frame_blocks * RT_FRAME_COUNT_STRIDE * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_frame_debug_timeline(rt_frame* f, int arm, unsigned long long* out /* [64][6] + [64] when reading */)
{
    if (!f) return fail(nullptr, "rt_frame_debug_timeline: frame is NULL");
    rt_ctx* ctx = f->ctx;
    (void)hipSetDevice(ctx->device);
    if (join_pipes(f) != RT_OK) return RT_ERROR;
    DCounters* c = f->ps[0].counters;
    if (arm)
    {
        HIPCHK(ctx, hipMemsetAsync(c->tl_start, 0xFF, sizeof(c->tl_start) + sizeof(c->tl_dry), ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(c->tl_end, 0, sizeof(c->tl_end) + sizeof(c->tl_ray_steps) + sizeof(c->tl_ray_ticks) + sizeof(c->tl_exit_hist),
            ctx->stream));
        f->timeline = 1;
        return RT_OK;
    }
    if (!out) return fail(ctx, "rt_frame_debug_timeline: NULL output");
    f->timeline = 0;
    std::vector<unsigned long long> h(384);
    HIPCHK(ctx, hipMemcpyAsync(h.data(), c->tl_start, 384 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int b = 0; b < 64; ++b)
    {
        bool ran = h[128 + b] != 0ull;
        out[b * 6 + 0] = ran ? h[b] : 0ull;
        out[b * 6 + 1] = ran && h[64 + b] != ~0ull ? h[64 + b] : 0ull;
        out[b * 6 + 2] = h[128 + b];
        out[b * 6 + 3] = h[192 + b];                    // most steps any ray took
        out[b * 6 + 4] = h[256 + b] >> 24;              // the slowest ray: ticks from hand-out to retirement ...
        out[b * 6 + 5] = h[256 + b] & 0xFFFFFFull;      // ... and its steps
    }
    for (int b = 0; b < 64; ++b) out[384 + b] = h[320 + b];   // waves that left in the b-th 25 us after the queue ran dry
    return RT_OK;
}

#include "debug_exports_impl.h"

int rt_debug_eval(rt_ctx* ctx, int fn, const float* a, const float* b, float* out, uint32_t n)
{
    if (!ctx || !a || !out) return fail(ctx, "rt_debug_eval: NULL argument");
    (void)hipSetDevice(ctx->device);
    float *da = nullptr, *db = nullptr, *dout = nullptr;
    struct Free { float*& a; float*& b; float*& c; ~Free() { for (float* p : {a, b, c}) if (p) (void)hipFree(p); } } guard{da, db, dout};
    HIPCHK(ctx, hipMalloc((void**)&da, (size_t)n * 4 + 16));
    HIPCHK(ctx, hipMalloc((void**)&dout, (size_t)n * 4 + 16));
    HIPCHK(ctx, hipMemcpy(da, a, (size_t)n * 4, hipMemcpyHostToDevice));
    if (b)
    {
        HIPCHK(ctx, hipMalloc((void**)&db, (size_t)n * 4 + 16));
        HIPCHK(ctx, hipMemcpy(db, b, (size_t)n * 4, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(k_debug_eval, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, fn, da, db, dout, n);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpy(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost));
    return RT_OK;
}

} // extern "C"

#include "group_impl.h"
