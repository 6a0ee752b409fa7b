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
#include "tree_select.h"
#include "tree_rotate.h"
#include <chrono>
#include "device_fold.h"

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
// ---- 4-wide quantized BVH (k_trace_w4) --------------------------------------
// A connected piece of the reference BVH2 (LinearBVHNode[], bvh.cpp:223-245) -- a node and up to two more interior
// nodes below it -- is folded into one 64-byte record: up to four "slots" = the frontier of that piece (which
// frontier: `collapse`, below).  Slot boxes are stored as 8-bit grid coordinates relative to a per-node frame
// (origin, power-of-two cell size per axis), rounded OUTWARD.
//
// Why results stay bit-identical to the reference ("wide traversal"):
//  * the frame is chosen so that origin + q * cell is exactly representable in binary32 for every
//    q in 0..255 (origin is a multiple of the cell, |origin| / cell < 2^23), so the kernel
//    dequantises WITHOUT rounding and evaluates the reference's own expression
//    fl(fl(b - o) * inv) on a box that contains the true one; that expression is monotone in b,
//    hence "true box passes  =>  stored box passes": interior culling only ever visits MORE;
//  * every leaf is box-tested again with its exact fp32 bounds and the ray's current t_max when it
//    is reached (the bounds travel in the leaf's first triangle record), and leaves are reached in
//    the reference's depth-first near/far order (the slots are brought into that order per direction octant by
//    tabulated exchanges, see `arrange`).  Node bounds are exact unions of their children's
//    (bvh.hpp:73, checked below), so a leaf's box passing implies that all its ancestors' boxes
//    pass: the reference tests the triangles of a leaf iff that leaf's own box test passes at that
//    point of the traversal -- which is exactly what the kernel evaluates.
// Record: q0 = (origin.xyz, meta)   meta = ex | ey << 8 | ez << 16 | occupied slots << 24 (biased exponents of the cell sizes)
//         q1 = (lo.x, lo.y, lo.z, hi.x)    one byte per slot in every dword
//         q2 = (hi.y, hi.z, ref0, ref1)    ref = wide node index | RT_LEAF_BIT + first triangle | RT_EMPTY_REF
//         q3 = (ref2, ref3, order, -)       order: for each of the 8 direction-sign octants o (bit a set = direction negative
//                                          along axis a) four bits at 4 * o = the conditional exchanges of slots (0,1), (2,3),
//                                          (0,2), (1,3), made in that order, that bring the occupied slots into the
//                                          reference's visit order (see `arrange` in build_wide_bvh)
// (struct WideNode: fold_kernels.h -- the device builds the same records, device_fold.h)

// Which BVH2 nodes become the four slots of a record (`collapse`):
//  RT_WIDE_TWO_LEVELS  the grandchildren (a child that is a leaf fills one slot): round 2's rule;
//  RT_WIDE_SAH         the frontier that minimises the expected number of wide-node visits: a record rooted at BVH2 node n
//                      is visited when a ray passes n's slot box (probability ~ area(n)), the interior nodes between n and
//                      its slots are never tested at all, leaves are what they are -- so the cost of a collapse is the sum of
//                      area(root) over its records, minimised exactly by a small dynamic programme over (node, slots to
//                      spend) (Ylitie, Karras, Laine 2017, section 4.1, for 4 slots and with the reference's leaves kept).
//                      The frontier of a record is then any of the five binary-tree shapes with four leaves (or fewer slots).
// The visit order of the slots stays the reference's for every shape: depth-first over the folded BVH2 nodes, the second
// child first where the ray is negative along that node's split axis (trace_bvh.cl:181-190).
enum { RT_WIDE_TWO_LEVELS = 0, RT_WIDE_SAH = 1 };

// false: the tree does not qualify (non-finite or non-nested bounds, child order): k_trace2 is used
bool build_wide_bvh(const rt_bvh_node* nodes, uint32_t nn, int collapse, std::vector<WideNode>& out, uint32_t& entry_ref,
    std::vector<uint32_t>* roots = nullptr /* the BVH2 node each record folds (tests) */,
    const ownbvh::Metric* metric = nullptr /* what "area" means for the SAH collapse (own_bvh.h); nullptr = surface area */,
    const double* weights = nullptr /* per BVH2 node: replaces the area altogether (a MEASURED visit frequency: FoldAdapt) */,
    const std::atomic<bool>* cancel = nullptr /* set by another thread: give up (false) at the next check -- a scene uploaded again does not wait */)
{
    auto cancelled = [&]() { return cancel && cancel->load(std::memory_order_relaxed); };
    auto is_leaf = [&](uint32_t i) { return (nodes[i].num_primitives_axis >> 16) != 0; };
    out.clear();
    if (is_leaf(0)) { entry_ref = RT_LEAF_BIT | nodes[0].offset; return true; }
    auto finite3 = [](const rt_float3& v) { return std::isfinite(v.x) && std::isfinite(v.y) && std::isfinite(v.z); };
    // pass 0: bounds finite and exactly nested (child inside parent), children after their parent
    for (uint32_t i = 0; i < nn; ++i)
    {
        const rt_bvh_node& n = nodes[i];
        if (!finite3(n.bounds_min) || !finite3(n.bounds_max)) return false;
        if (is_leaf(i)) continue;
        if (i + 1 >= nn || n.offset >= nn || n.offset <= i + 1 || (n.num_primitives_axis & 0xFFFFu) > 2u) return false;
        for (uint32_t c : {i + 1, n.offset})
        {
            const rt_bvh_node& k = nodes[c];
            if (k.bounds_min.x < n.bounds_min.x || k.bounds_min.y < n.bounds_min.y || k.bounds_min.z < n.bounds_min.z ||
                k.bounds_max.x > n.bounds_max.x || k.bounds_max.y > n.bounds_max.y || k.bounds_max.z > n.bounds_max.z)
                return false;
        }
    }
    // the collapse: split[n][k] = slots given to n's first child when n is folded with k slots to spend (k = 2..4, the second
    // child gets the rest); a child with i >= 2 slots is folded too iff open[c] has bit i set, otherwise it is one slot
    std::vector<uint8_t> split((size_t)nn * 5u, 0), open(nn, 0);
    if (collapse == RT_WIDE_SAH)
    {
        // T[n] = cost of the best collapse of n's subtree with a record rooted at n; F[n][k] = the same without the root's own
        // visit, n's subtree covered by k slots.  Children have larger indices than their parent (pass 0): one backward sweep.
        std::vector<double> T(nn, 0.0), F((size_t)nn * 5u, 0.0);
        auto G = [&](uint32_t c, uint32_t i) { return is_leaf(c) ? 0.0 : (i >= 2u ? std::min(T[c], F[(size_t)c * 5u + i]) : T[c]); };
        for (uint32_t n = nn; n-- > 0;)
        {
            if ((n & 0xFFFFu) == 0u && cancelled()) return false;
            if (is_leaf(n)) continue;
            const uint32_t l = n + 1, r = nodes[n].offset;
            for (uint32_t k = 2; k <= 4; ++k)
            {
                double best = 0.0; uint32_t at = 0;
                for (uint32_t i = 1; i < k; ++i)
                {
                    const double c = G(l, i) + G(r, k - i);
                    if (at == 0 || c < best) { best = c; at = i; }
                }
                F[(size_t)n * 5u + k] = best;
                split[(size_t)n * 5u + k] = (uint8_t)at;
            }
            const rt_bvh_node& b = nodes[n];
            const double dx = (double)b.bounds_max.x - b.bounds_min.x, dy = (double)b.bounds_max.y - b.bounds_min.y,
                         dz = (double)b.bounds_max.z - b.bounds_min.z;
            const float bmn[3] = {b.bounds_min.x, b.bounds_min.y, b.bounds_min.z}, bmx[3] = {b.bounds_max.x, b.bounds_max.y, b.bounds_max.z};
            T[n] = (weights ? weights[n] : metric ? metric->of(bmn, bmx) : dx * dy + dy * dz + dz * dx) + F[(size_t)n * 5u + 4u];
            for (uint32_t i = 2; i <= 4; ++i)
                if (F[(size_t)n * 5u + i] < T[n]) open[n] |= (uint8_t)(1u << i);
        }
    }
    else
    {
        for (uint32_t n = 0; n < nn; ++n)
        {
            if (is_leaf(n)) continue;
            split[(size_t)n * 5u + 4u] = 2; split[(size_t)n * 5u + 3u] = 2; split[(size_t)n * 5u + 2u] = 1;
            open[n] = 1u << 2;                                              // a child with two slots to spend shows its children
        }
    }
    // One record: its slots in the BVH2's depth-first order and, per direction-sign octant, the positions in visit order.
    struct Fold { uint32_t slot[4]; uint32_t n_slots; uint8_t visit[8][4]; };
    struct Local
    {
        const rt_bvh_node* nodes; const std::vector<uint8_t>& split; const std::vector<uint8_t>& open;
        bool leaf(uint32_t i) const { return (nodes[i].num_primitives_axis >> 16) != 0; }
        // n folded with k slots to spend: appends n's slots to `f` and returns, per octant, their positions in visit order
        void fold(uint32_t n, uint32_t k, Fold& f, uint8_t (&visit)[8][4], uint32_t& count) const
        {
            const uint32_t c[2] = {n + 1, nodes[n].offset};
            const uint32_t give[2] = {split[(size_t)n * 5u + k], k - split[(size_t)n * 5u + k]};
            uint8_t part[2][8][4];
            uint32_t len[2] = {0, 0};
            for (int i = 0; i < 2; ++i)
            {
                if (!leaf(c[i]) && give[i] >= 2u && ((open[c[i]] >> give[i]) & 1u)) fold(c[i], give[i], f, part[i], len[i]);
                else
                {
                    for (int o = 0; o < 8; ++o) part[i][o][0] = (uint8_t)f.n_slots;
                    f.slot[f.n_slots++] = c[i];
                    len[i] = 1;
                }
            }
            const uint32_t axis = nodes[n].num_primitives_axis & 0xFFFFu;
            for (uint32_t o = 0; o < 8; ++o)
            {
                // trace_bvh.cl:181-190: the near child is the second one when the ray is negative along the split axis
                const int first = (int)((o >> axis) & 1u);
                uint32_t at = 0;
                for (uint32_t j = 0; j < len[first]; ++j) visit[o][at++] = part[first][o][j];
                for (uint32_t j = 0; j < len[first ^ 1]; ++j) visit[o][at++] = part[first ^ 1][o][j];
            }
            count = len[0] + len[1];
        }
    } local{nodes, split, open};
    auto fold_of = [&](uint32_t n, Fold& f)
    {
        f.n_slots = 0;
        for (int k = 0; k < 4; ++k) f.slot[k] = RT_EMPTY_REF;
        uint32_t count = 0;
        local.fold(n, 4u, f, f.visit, count);
    };
    // Where the slots of a record are stored.  The kernel brings them into visit order with FOUR conditional exchanges --
    // (0,1), (2,3), (0,2), (1,3), one table bit each per direction octant: two instructions more than the three decisions
    // round 2 tabulated for the one shape it folded -- and that network does not realise every permutation; but for each of
    // the five shapes (and their smaller relatives) there is a placement of the slots for which it realises all the orders
    // the shape can ask for (exhaustive search: tests/test_wide_bvh.py).  Found here by trying the 24 placements, once per
    // distinct (slot count, eight visit orders); depth-first order is tried first, which is what the balanced shape keeps.
    struct Arrangement { uint8_t place[4]; uint32_t order; };
    typedef std::map<std::array<uint8_t, 33>, Arrangement> ArrangementCache;
    auto arrange = [&](const Fold& f, ArrangementCache& arrangements) -> const Arrangement*
    {
        std::array<uint8_t, 33> key{};
        key[0] = (uint8_t)f.n_slots;
        for (int o = 0; o < 8; ++o)
            for (uint32_t j = 0; j < f.n_slots; ++j) key[1 + 4 * o + j] = f.visit[o][j];
        auto it = arrangements.find(key);
        if (it != arrangements.end()) return &it->second;
        uint8_t place[4] = {0, 1, 2, 3};                                   // place[j] = slot position of the j-th node in depth-first order
        do
        {
            uint8_t node_at[4] = {255, 255, 255, 255};
            for (uint32_t j = 0; j < f.n_slots; ++j) node_at[place[j]] = (uint8_t)j;
            Arrangement a{};
            bool all = true;
            for (uint32_t o = 0; o < 8 && all; ++o)
            {
                bool found = false;
                for (uint32_t bits = 0; bits < 16u && !found; ++bits)
                {
                    uint8_t pos[4] = {0, 1, 2, 3};
                    static const int ex[4][2] = {{0, 1}, {2, 3}, {0, 2}, {1, 3}};
                    for (int c = 0; c < 4; ++c)
                        if ((bits >> c) & 1u) std::swap(pos[ex[c][0]], pos[ex[c][1]]);
                    // the occupied slots, in the order the kernel will look at them, must be the reference's visit order
                    uint32_t at = 0;
                    bool same = true;
                    for (int k = 0; k < 4 && same; ++k)
                        if (node_at[pos[k]] != 255) same = node_at[pos[k]] == f.visit[o][at++];
                    if (same) { a.order |= bits << (4u * o); found = true; }
                }
                all = found;
            }
            if (all)
            {
                memcpy(a.place, place, 4);
                return &arrangements.emplace(key, a).first->second;
            }
        } while (std::next_permutation(place, place + 4));
        return nullptr;
    };
    // pass 1: wide nodes in depth-first order (slot 0's subtree first), like the reference's flattening
    std::vector<uint32_t> wide_of(nn, RT_EMPTY_REF), todo, order, depth_of;
    todo.push_back(0);
    depth_of.push_back(1);
    while (!todo.empty())
    {
        uint32_t n = todo.back(), depth = depth_of.back();
        todo.pop_back();
        depth_of.pop_back();
        if ((order.size() & 0xFFFFu) == 0u && cancelled()) return false;
        if (depth > 33u) return false;                                     // <= 3 pending slots per level must fit RT_W4_STACK_MAX
        // a node reached twice (several parents share a child) is not a tree: the walk below would append once per PATH
        if (wide_of[n] != RT_EMPTY_REF || order.size() >= nn) return false;
        wide_of[n] = (uint32_t)order.size();
        order.push_back(n);
        Fold f;
        fold_of(n, f);
        for (int k = 3; k >= 0; --k)
            if (f.slot[k] != RT_EMPTY_REF && !is_leaf(f.slot[k])) { todo.push_back(f.slot[k]); depth_of.push_back(depth + 1u); }
    }
    if (order.size() >= (1u << 26)) return false;                          // 32-bit byte offsets in the kernel
    // pass 2: records (independent of each other: host threads, each with its own cache of arrangements)
    out.resize(order.size());
    auto make_record = [&](size_t w, ArrangementCache& cache) -> bool
    {
        const uint32_t n = order[w];
        Fold f;
        fold_of(n, f);
        const Arrangement* arr = arrange(f, cache);
        if (!arr) return false;                                            // cannot happen (every shape has an arrangement: tests/test_wide_bvh.py)
        uint32_t slot[4] = {RT_EMPTY_REF, RT_EMPTY_REF, RT_EMPTY_REF, RT_EMPTY_REF};
        for (uint32_t j = 0; j < f.n_slots; ++j) slot[arr->place[j]] = f.slot[j];
        WideNode& r = out[w];
        memset(&r, 0, sizeof(r));
        const float nmin[3] = {nodes[n].bounds_min.x, nodes[n].bounds_min.y, nodes[n].bounds_min.z};
        const float nmax[3] = {nodes[n].bounds_max.x, nodes[n].bounds_max.y, nodes[n].bounds_max.z};
        float origin[3];
        int exps[3];
        for (int a = 0; a < 3; ++a)
        {
            // cell = 2^e: 254 cells span the node (one spare for the floor of the origin), and the grid
            // stays exactly representable: |origin| / cell < 2^23 leaves room for + 255 below 2^24
            const double extent = (double)nmax[a] - (double)nmin[a];
            const double amax = std::max(std::fabs((double)nmin[a]), std::fabs((double)nmax[a]));
            int e = -126;
            if (extent > 0.0) e = std::max(e, (int)std::ceil(std::log2(extent / 254.0)));
            while (std::ldexp(254.0, e) < extent) ++e;
            while (amax > 0.0 && amax / std::ldexp(1.0, e) >= 8388608.0 - 256.0) ++e;
            // k_trace_w4 evaluates slab distances as q * (cell * inv) + (origin - org) * inv: bounded operands keep that
            // finite for every ray it accepts (trace_kernels.h, loop C)
            if (e > 20 || amax >= 268435456.0) return false;
            const double cell = std::ldexp(1.0, e);
            const double o = std::floor((double)nmin[a] / cell) * cell;
            origin[a] = (float)o;
            if ((double)origin[a] != o) return false;                     // cannot happen by construction
            exps[a] = e;
        }
        r.ox = origin[0]; r.oy = origin[1]; r.oz = origin[2];
        r.meta = (uint32_t)(exps[0] + 127) | (uint32_t)(exps[1] + 127) << 8 | (uint32_t)(exps[2] + 127) << 16 | f.n_slots << 24;
        r.order = arr->order;
        for (int k = 0; k < 4; ++k)
        {
            if (slot[k] == RT_EMPTY_REF)
            {
                r.ref[k] = RT_EMPTY_REF;
                for (int a = 0; a < 3; ++a) { r.lo[a] |= 255u << (8 * k); }   // lo 255 > hi 0: never hit
                continue;
            }
            const rt_bvh_node& c = nodes[slot[k]];
            r.ref[k] = is_leaf(slot[k]) ? (RT_LEAF_BIT | c.offset) : wide_of[slot[k]];
            const float cmin[3] = {c.bounds_min.x, c.bounds_min.y, c.bounds_min.z};
            const float cmax[3] = {c.bounds_max.x, c.bounds_max.y, c.bounds_max.z};
            for (int a = 0; a < 3; ++a)
            {
                const double cell = std::ldexp(1.0, exps[a]);
                double lo = std::floor(((double)cmin[a] - (double)origin[a]) / cell);
                double hi = std::ceil(((double)cmax[a] - (double)origin[a]) / cell);
                // the difference above is rounded (a bound of 1e-17 beside an origin of -0.2 vanishes in it): settle the
                // containment on the grid points themselves, which are exact in binary32 and binary64 alike
                while ((double)origin[a] + lo * cell > (double)cmin[a]) lo -= 1.0;
                while ((double)origin[a] + hi * cell < (double)cmax[a]) hi += 1.0;
                if (lo < 0.0 || hi > 255.0 || lo > hi) return false;      // cannot happen: the child is inside the node
                r.lo[a] |= (uint32_t)lo << (8 * k);
                r.hi[a] |= (uint32_t)hi << (8 * k);
            }
        }
        return true;
    };
    {
        const size_t n_records = order.size();
        // (a fold with measured weights is an adaptation's, made beside the render loop: 16 threads, adapt_threads below)
        const unsigned n_threads = (unsigned)std::min<size_t>(std::max(1u, std::min(std::thread::hardware_concurrency(), weights ? 16u : 32u)), n_records / 4096 + 1);
        std::atomic<bool> ok{true};
        auto run = [&](size_t w0, size_t w1)
        {
            ArrangementCache cache;
            for (size_t w = w0; w < w1 && ok.load(std::memory_order_relaxed); ++w)
                if (((w & 0x3FFFu) == 0u && cancelled()) || !make_record(w, cache)) ok.store(false);
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < n_threads; ++t) pool.emplace_back(run, n_records * t / n_threads, n_records * (t + 1) / n_threads);
        run(0, n_records / n_threads);
        for (auto& th : pool) th.join();
        if (!ok) return false;
    }
    entry_ref = 0;
    if (roots) *roots = order;
    return true;
}

// ---- RT_CTX_OPT_WIDE_LAYOUT = 1: the records in PAIRS (round 6) -------------------------------------------------------------------------
// The L2 of gfx950 fetches 128-byte lines (every read request of the traversal kernels at the fabric is a 128-byte one: TCC_EA0_RDREQ_128B,
// profiles/r06_fetch_size_calibration.json), so a 64-byte record that misses brings its line-mate along whether anybody wants it or not.  In the fold's
// own order -- depth first -- the line-mate of a record at an even index is its first slot's record and that of one at an odd index is whatever came
// before it.  Here the line-mate is CHOSEN: every record that has interior slots is stored at an even index with the child it hands most rays on to right
// behind it (by the weight the fold was made for: the measured crossings of an adaptation, else the area of the child's box), so a visit of that child
// never misses after the visit of its parent that must precede it.  A pure permutation of the records (refs are indices): no result depends on it.
// weight(record) -> the visit weight of the record's root box.
template <class W>
void pair_layout(std::vector<WideNode>& wide, std::vector<uint32_t>* roots, W&& weight)
{
    const uint32_t n = (uint32_t)wide.size();
    if (n < 3u) return;
    auto interior = [](uint32_t ref) { return ref != RT_EMPTY_REF && !(ref & RT_LEAF_BIT); };
    std::vector<uint32_t> order, singles, todo;
    std::vector<uint8_t> placed(n, 0);
    order.reserve(n);
    todo.push_back(0u);
    while (!todo.empty())
    {
        const uint32_t r = todo.back();
        todo.pop_back();
        if (r >= n || placed[r]) continue;
        placed[r] = 1;
        uint32_t best = RT_EMPTY_REF;
        double best_w = -1.0;
        for (uint32_t ref : wide[r].ref)
            if (interior(ref) && ref < n && !placed[ref]) { const double w = weight(ref); if (best == RT_EMPTY_REF || w > best_w) { best = ref; best_w = w; } }
        if (best == RT_EMPTY_REF) { singles.push_back(r); continue; }
        placed[best] = 1;
        order.push_back(r); order.push_back(best);
        // what hangs below the two, depth first (the head's other children before the tail's: they are the nearer relatives)
        for (int k = 3; k >= 0; --k) { const uint32_t ref = wide[best].ref[k]; if (interior(ref) && ref < n && !placed[ref]) todo.push_back(ref); }
        for (int k = 3; k >= 0; --k) { const uint32_t ref = wide[r].ref[k]; if (interior(ref) && ref < n && !placed[ref]) todo.push_back(ref); }
    }
    order.insert(order.end(), singles.begin(), singles.end());
    if (order.size() != n || order[0] != 0u) return;                       // (not a tree over all records: leave it as it is)
    std::vector<uint32_t> at(n);
    for (uint32_t i = 0; i < n; ++i) at[order[i]] = i;
    std::vector<WideNode> out(n);
    for (uint32_t i = 0; i < n; ++i)
    {
        WideNode r = wide[order[i]];
        for (uint32_t& ref : r.ref) if (interior(ref) && ref < n) ref = at[ref];
        out[i] = r;
    }
    wide.swap(out);
    if (roots && roots->size() == n)
    {
        std::vector<uint32_t> rn(n);
        for (uint32_t i = 0; i < n; ++i) rn[i] = (*roots)[order[i]];
        roots->swap(rn);
    }
}

// the static folds' weight: the area (the own trees': their metric) of the box a record tests
template <class M>
void pair_layout_by_area(std::vector<WideNode>& wide, std::vector<uint32_t>& roots, const rt_bvh_node* nodes, uint32_t nn, const M* metric)
{
    if (roots.size() != wide.size()) return;
    const std::vector<uint32_t> r0 = roots;                                // (weights are asked for by OLD record index while `roots` is being permuted at the end only)
    pair_layout(wide, &roots, [&](uint32_t rec) -> double
    {
        const uint32_t node = r0[rec];
        if (node >= nn) return 0.0;
        const rt_bvh_node& b = nodes[node];
        const float mn[3] = {b.bounds_min.x, b.bounds_min.y, b.bounds_min.z}, mx[3] = {b.bounds_max.x, b.bounds_max.y, b.bounds_max.z};
        if (metric) return metric->of(mn, mx);
        const double dx = (double)mx[0] - mn[0], dy = (double)mx[1] - mn[1], dz = (double)mx[2] - mn[2];
        return dx * dy + dy * dz + dz * dx;
    });
}

// The ray population a shadow tree serves (own_bvh.h): shadow rays go to the analytic lights only (hit_surface.cl:114-146,
// light.h:30-65), one uniformly chosen per hit -- towards a directional light they all share its direction, towards a point
// light they come from everywhere.
ownbvh::Metric shadow_metric(const rt_light* lights, uint32_t n, double iso_share)
{
    ownbvh::Metric m;
    m.iso = 0.0;
    for (uint32_t i = 0; i < n; ++i)
    {
        const rt_light& l = lights[i];
        const double len = std::sqrt((double)l.origin.x * l.origin.x + (double)l.origin.y * l.origin.y + (double)l.origin.z * l.origin.z);
        if (l.type == RT_LIGHT_TYPE_POINT || !(len > 0.0) || !std::isfinite(len)) { m.iso += 1.0; continue; }
        m.dirs.push_back({std::fabs(l.origin.x / len), std::fabs(l.origin.y / len), std::fabs(l.origin.z / len)});
    }
    if (m.dirs.empty()) m.iso = 1.0;
    else m.iso += iso_share * (double)m.dirs.size();   // some isotropy keeps boxes that are thin along d from growing without bound
    return m;
}

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
        worker = std::thread([this, sd, m]()
        {
            const auto t0 = std::chrono::steady_clock::now();
            ok = ownbvh::build(sd->nodes, sd->num_nodes, m, bvh2);
            build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            ok = ok && fold_it(m);
        });
    }
    void join() { if (worker.joinable()) worker.join(); }
    ~OwnTree() { join(); if (d_wide) (void)hipFree(d_wide); }
};

// true: walk the own tree; false: keep the reference topology
bool choose_tree(const rt_scene_desc* sd, const std::vector<WideNode>& ref_wide, uint32_t ref_entry, bool shadow, uint32_t mode,
    OwnTree& own, std::string& report)
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
    std::vector<uint32_t> leaf_of_first(nt, 0u);
    for (uint32_t i = 0; i < nn; ++i)
        if ((sd->nodes[i].num_primitives_axis >> 16) != 0 && sd->nodes[i].offset < nt) leaf_of_first[sd->nodes[i].offset] = i;
    const std::vector<treesel::ProxyRay> rays = treesel::proxy_rays(sd->triangles, nt, sd->lights, sd->num_lights, 8192u, shadow);
    if (rays.empty()) { report += shadow ? "shadow tree: no lights, nothing to measure -> reference topology\n" : "closest-hit tree: nothing to measure -> reference topology\n"; return false; }
    const double c_ref = treesel::walk_cost((const treesel::Record*)ref_wide.data(), (uint32_t)ref_wide.size(), ref_entry, sd->nodes, leaf_of_first.data(), sd->triangles, rays, shadow);
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
    bool device_fold = false;                                            // RT_CTX_OPT_DEVICE_FOLD: crossing counts and re-folds on `device`
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

// Host threads one side of an adaptation may use beside the render loop: the closest-hit and the shadow side run together, one process per
// GPU runs one context each, so 16 + 16 threads x 8 ranks stays within a 256-core host (2 x 32 per context oversubscribed it).
static unsigned adapt_threads(size_t work_items, size_t per_thread)
{
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    return (unsigned)std::min<size_t>(std::min(hw, 16u), work_items / per_thread + 1);
}
static std::atomic<uint64_t> g_truncated_walks{0};   // host walks (weights only) that met a binary tree deeper than their 126-entry stack

// counts[n] = rays whose slab test of binary-tree node n passes within [0, o.w] (plain binary32 arithmetic: a weight, not a result)
void count_box_passes(const rt_bvh_node* nodes, uint32_t nn, const float4* o, const float4* d, size_t n_rays, std::vector<uint32_t>& counts,
    const std::atomic<bool>& cancel)
{
    counts.assign(nn, 0u);
    auto run = [&](size_t r0, size_t r1)
    {
        uint32_t stack[128];
        for (size_t r = r0; r < r1 && !cancel.load(std::memory_order_relaxed); ++r)
        {
            const float org[3] = {o[r].x, o[r].y, o[r].z}, inv[3] = {1.0f / d[r].x, 1.0f / d[r].y, 1.0f / d[r].z};
            const float t_max = o[r].w;
            int sp = 0;
            stack[sp++] = 0;
            while (sp > 0)
            {
                const uint32_t n = stack[--sp];
                const rt_bvh_node& b = nodes[n];
                const float mn[3] = {b.bounds_min.x, b.bounds_min.y, b.bounds_min.z}, mx[3] = {b.bounds_max.x, b.bounds_max.y, b.bounds_max.z};
                float t0 = 0.0f, t1 = t_max;
                for (int a = 0; a < 3; ++a)
                {
                    const float ta = (mn[a] - org[a]) * inv[a], tb = (mx[a] - org[a]) * inv[a];
                    t0 = std::fmax(t0, std::fmin(ta, tb));         // fmin / fmax drop a NaN (0 * inf): conservative, like the kernels
                    t1 = std::fmin(t1, std::fmax(ta, tb));
                }
                if (!(t0 <= t1)) continue;
                __atomic_fetch_add(&counts[n], 1u, __ATOMIC_RELAXED);
                if ((b.num_primitives_axis >> 16) != 0) continue;
                if (sp > 125) { g_truncated_walks.fetch_add(1, std::memory_order_relaxed); continue; }
                if (b.offset >= nn || n + 1u >= nn) continue;
                stack[sp++] = b.offset;
                stack[sp++] = n + 1u;
            }
        }
    };
    const unsigned n_threads = adapt_threads(n_rays, 2048);
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < n_threads; ++t) pool.emplace_back(run, n_rays * t / n_threads, n_rays * (t + 1) / n_threads);
    run(0, n_rays / n_threads);
    for (auto& th : pool) th.join();
}

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
            if (truncated) g_truncated_walks.fetch_add(truncated, std::memory_order_relaxed);
        }
        if (!on_device) (void)hipGetLastError();
    }
    if (!on_device) count_box_passes(tree.data(), nn, o.data(), d.data(), o.size(), counts, cancel);
    if (cancel.load()) return false;
    // the measured passes, plus a twentieth of their sum spread by surface area: boxes no probe ray met still fold sensibly
    std::vector<double> w(nn);
    double total = 0.0, area_sum = 0.0;
    for (uint32_t n = 0; n < nn; ++n)
    {
        const rt_bvh_node& b = tree[n];
        const double dx = (double)b.bounds_max.x - b.bounds_min.x, dy = (double)b.bounds_max.y - b.bounds_min.y, dz = (double)b.bounds_max.z - b.bounds_min.z;
        w[n] = dx * dy + dy * dz + dz * dx;
        area_sum += w[n];
        total += (double)counts[n];
    }
    if (!(total > 0.0) || !(area_sum > 0.0) || !std::isfinite(area_sum)) return false;
    const double prior = 0.05 * total / area_sum;
    for (uint32_t n = 0; n < nn; ++n) w[n] = (double)counts[n] + prior * w[n];
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
        const std::vector<uint32_t> r0 = roots_new;
        pair_layout(out, &roots_new, [&](uint32_t rec) { return r0[rec] < nn ? w[r0[rec]] : 0.0; });
    }
    if (roots_out) roots_out->swap(roots_new);
    return cost[1] < cost[0];
}

// Mode bit 4: the ORDER in which a shadow ray looks at the slots of a record is free -- its verdict is an OR over the leaves it reaches -- and
// k_trace_w4<shadow> takes them as they are stored (the exchange network is the closest-hit rays': trace_kernels.h, w4_test_slots).  An occluded
// ray stops at its first hit, so each record's slots are stored likeliest occluder first: by how many probe shadow rays had their NEAREST occluder
// in the slot's subtree.  (tools/fold_weight_study.py --order: - 12 % steps per shadow ray on the headline scene with a quarter of them
// occluded, - 25 % for the occluded ones.)  The nearest occluder of a probe ray is found here, on the host: a plain closest-hit walk of the
// reference's binary tree with Moeller-Trumbore in binary32 -- a statistic, not a result.
void nearest_occluders(const std::vector<rt_bvh_node>& tree, const std::vector<float>& tri9, const std::vector<float4>& o, const std::vector<float4>& d,
    std::vector<uint32_t>& prim, const std::atomic<bool>& cancel)
{
    const size_t n_rays = o.size();
    const uint32_t nn = (uint32_t)tree.size(), nt = (uint32_t)(tri9.size() / 9);
    prim.assign(n_rays, RT_INVALID_ID);
    auto run = [&](size_t r0, size_t r1)
    {
        uint32_t stack[128];
        for (size_t r = r0; r < r1 && !cancel.load(std::memory_order_relaxed); ++r)
        {
            const float org[3] = {o[r].x, o[r].y, o[r].z}, dir[3] = {d[r].x, d[r].y, d[r].z}, inv[3] = {1.0f / d[r].x, 1.0f / d[r].y, 1.0f / d[r].z};
            float t_max = o[r].w;
            int sp = 0;
            stack[sp++] = 0;
            while (sp > 0)
            {
                const uint32_t n = stack[--sp];
                const rt_bvh_node& b = tree[n];
                const float mn[3] = {b.bounds_min.x, b.bounds_min.y, b.bounds_min.z}, mx[3] = {b.bounds_max.x, b.bounds_max.y, b.bounds_max.z};
                float t0 = 0.0f, t1 = t_max;
                for (int a = 0; a < 3; ++a)
                {
                    const float ta = (mn[a] - org[a]) * inv[a], tb = (mx[a] - org[a]) * inv[a];
                    t0 = std::fmax(t0, std::fmin(ta, tb));
                    t1 = std::fmin(t1, std::fmax(ta, tb));
                }
                if (!(t0 <= t1)) continue;
                const uint32_t count = b.num_primitives_axis >> 16;
                if (count != 0)
                {
                    for (uint32_t k = 0; k < count && b.offset + k < nt; ++k)
                    {
                        const float* p = &tri9[(size_t)(b.offset + k) * 9];
                        const float e1[3] = {p[3] - p[0], p[4] - p[1], p[5] - p[2]}, e2[3] = {p[6] - p[0], p[7] - p[1], p[8] - p[2]};
                        const float pv[3] = {dir[1] * e2[2] - dir[2] * e2[1], dir[2] * e2[0] - dir[0] * e2[2], dir[0] * e2[1] - dir[1] * e2[0]};
                        const float det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
                        if (!(std::fabs(det) > 1e-8f)) continue;
                        const float id = 1.0f / det;
                        const float tv[3] = {org[0] - p[0], org[1] - p[1], org[2] - p[2]};
                        const float u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) * id;
                        if (!(u >= 0.0f && u <= 1.0f)) continue;
                        const float qv[3] = {tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0]};
                        const float v = (dir[0] * qv[0] + dir[1] * qv[1] + dir[2] * qv[2]) * id;
                        if (!(v >= 0.0f && u + v <= 1.0f)) continue;
                        const float t = (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]) * id;
                        if (t > 0.0f && t < t_max) { t_max = t; prim[r] = b.offset + k; }
                    }
                    continue;
                }
                if (sp > 125) { g_truncated_walks.fetch_add(1, std::memory_order_relaxed); continue; }
                if (b.offset >= nn || n + 1u >= nn) continue;
                stack[sp++] = b.offset;
                stack[sp++] = n + 1u;
            }
        }
    };
    const unsigned n_threads = adapt_threads(n_rays, 2048);
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < n_threads; ++t) pool.emplace_back(run, n_rays * t / n_threads, n_rays * (t + 1) / n_threads);
    run(0, n_rays / n_threads);
    for (auto& th : pool) th.join();
}

// The slots of every record of `wide` (a fold of `tree`, record w testing node roots[w]) stored by descending count of probe rays whose nearest
// occluder (prim[]) lies in the slot's subtree; equal counts keep their places.  A pure permutation within each record.  Returns the records changed.
uint32_t occluder_first(std::vector<WideNode>& wide, const std::vector<uint32_t>& roots, const std::vector<rt_bvh_node>& tree, const std::vector<uint32_t>& prim)
{
    const uint32_t nn = (uint32_t)tree.size();
    if (wide.empty() || roots.size() != wide.size() || nn == 0) return 0;
    std::vector<uint32_t> parent(nn, RT_EMPTY_REF), hit(nn, 0u);
    uint32_t max_prim = 0;
    for (uint32_t i = 0; i < nn; ++i)
    {
        const uint32_t count = tree[i].num_primitives_axis >> 16;
        if (count != 0) { max_prim = std::max(max_prim, tree[i].offset + count); continue; }
        if (i + 1u < nn) parent[i + 1u] = i;
        if (tree[i].offset < nn) parent[tree[i].offset] = i;
    }
    std::vector<uint32_t> leaf_of(max_prim, RT_EMPTY_REF);                 // primitive -> the leaf node of `tree` that holds it
    for (uint32_t i = 0; i < nn; ++i)
    {
        const uint32_t count = tree[i].num_primitives_axis >> 16;
        for (uint32_t k = 0; k < count; ++k) leaf_of[tree[i].offset + k] = i;
    }
    for (uint32_t p : prim)
    {
        if (p >= max_prim) continue;
        uint32_t guard = 0;
        for (uint32_t n = leaf_of[p]; n != RT_EMPTY_REF && guard < 256u; n = parent[n], ++guard) ++hit[n];
    }
    uint32_t changed = 0;
    for (size_t w = 0; w < wide.size(); ++w)
    {
        WideNode& r = wide[w];
        uint32_t score[4]; int idx[4] = {0, 1, 2, 3};
        bool any = false;
        for (int k = 0; k < 4; ++k)
        {
            const uint32_t ref = r.ref[k];
            uint32_t node = RT_EMPTY_REF;
            if (ref == RT_EMPTY_REF) { score[k] = 0; continue; }
            if (ref & RT_LEAF_BIT) { const uint32_t first = ref & ~RT_LEAF_BIT; node = first < max_prim ? leaf_of[first] : RT_EMPTY_REF; }
            else if (ref < roots.size()) node = roots[ref];
            score[k] = node < nn ? hit[node] + 1u : 1u;                    // occupied slots before empty ones
            any = true;
        }
        if (!any) continue;
        std::stable_sort(idx, idx + 4, [&](int x, int y) { return score[x] > score[y]; });
        if (idx[0] == 0 && idx[1] == 1 && idx[2] == 2 && idx[3] == 3) continue;
        WideNode q = r;
        for (int a = 0; a < 3; ++a) { q.lo[a] = 0; q.hi[a] = 0; }
        for (int k = 0; k < 4; ++k)
        {
            q.ref[k] = r.ref[idx[k]];
            for (int a = 0; a < 3; ++a)
            {
                q.lo[a] |= ((r.lo[a] >> (8 * idx[k])) & 0xFFu) << (8 * k);
                q.hi[a] |= ((r.hi[a] >> (8 * idx[k])) & 0xFFu) << (8 * k);
            }
        }
        r = q;
        ++changed;
    }
    return changed;
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
    const int fold_device = a->device_fold ? a->device : -1;
    bool ok = refold_for_rays(tree, a->sh_o, a->sh_d, roots, a->wide_sh, a->entry_sh, a->cost[1], a->cancel, &a->roots_sh_new, fold_device, a->pairs);
    if (!(a->mode.load() & 8u) || a->sh_o.empty() || a->cancel.load()) return ok;
    std::vector<rt_bvh_node> rotated;
    double crossings[2] = {0.0, 0.0};
    const uint32_t made = treerot::rotate(tree.data(), (uint32_t)tree.size(), (const float*)a->sh_o.data(), (const float*)a->sh_d.data(), a->sh_o.size(), 8, rotated, crossings, &a->cancel);
    if (made == 0 || rotated.size() != tree.size() || a->cancel.load()) return ok;
    std::vector<WideNode> wide;
    std::vector<uint32_t> roots_rot;
    uint32_t entry = 0;
    double cost[2] = {0.0, 0.0};
    const std::vector<uint32_t> top{0u};                                   // (the rotated tree has no current fold: only cost[1] is read)
    (void)refold_for_rays(rotated, a->sh_o, a->sh_d, top, wide, entry, cost, a->cancel, &roots_rot, fold_device, a->pairs);
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
    const bool ok = adapt_shadow_candidate(a);
    if (ok && (a->mode.load() & 16u) && !a->tri9.empty() && !a->wide_sh.empty() && !a->cancel.load())
    {
        // the candidate's slots, likeliest occluder first (mode bit 4)
        std::vector<uint32_t> prim;
        nearest_occluders(a->bvh2, a->tri9, a->sh_o, a->sh_d, prim, a->cancel);
        const std::vector<rt_bvh_node>& tree = a->rotations != 0 ? a->bvh2_sh_new : (a->bvh2_sh.empty() ? a->bvh2 : a->bvh2_sh);
        if (!a->cancel.load()) a->reordered = occluder_first(a->wide_sh, a->roots_sh_new, tree, prim);
    }
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
    probe_unpack(a);
    if (!a->o.empty())
    {
        std::thread shadow([a]() { a->ok_sh = adapt_shadow_side(a); });
        a->ok = refold_for_rays(a->bvh2, a->o, a->d, a->roots, a->wide, a->entry, a->cost[0], a->cancel, &a->roots_new, a->device_fold ? a->device : -1, a->pairs);
        shadow.join();
        fold_upload(a);
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
    OwnTree own_sh, own_cl;
    const bool may_own = ctx->build_wide == 1u && (uint64_t)nt * 64 <= 0xFFFFFFFFull && (sd->nodes[0].num_primitives_axis >> 16) == 0;
    if (ctx->device_fold && ctx->build_wide == 1u) { own_sh.device = ctx->device; own_cl.device = ctx->device; }
    own_sh.pairs = own_cl.pairs = ctx->wide_layout != 0u;
    if (may_own && ctx->shadow_tree) own_sh.start(sd, true, ctx->shadow_tree);
    if (may_own && ctx->closest_tree) own_cl.start(sd, false, ctx->closest_tree);

    // --- BVH re-layout: LinearBVHNode[] (bvh.cpp:223-245) -> child-pair records
    // Record order = cache-friendly "treelet" layout: a breadth-first cluster of up to
    // RT_TREELET_NODES interior nodes is stored contiguously (7 records = 448 B, i.e. the
    // next three levels below a node share a few cache lines), then the clusters hanging
    // off it, depth first.  This is a pure permutation of records: traversal decisions and
    // results do not depend on it.  (RT_TREELET_NODES = 1 gives the reference's DFS order.)
    std::vector<uint32_t> interior_index(nn, RT_EMPTY_REF);
    uint32_t n_interior = 0;
    {
        auto is_interior = [&](uint32_t i) { return (sd->nodes[i].num_primitives_axis >> 16) == 0; };
        const uint32_t kTreelet = ctx->treelet_nodes ? ctx->treelet_nodes : 1u;
        std::vector<uint32_t> roots, cluster, frontier;
        if (is_interior(0)) roots.push_back(0);
        while (!roots.empty())
        {
            uint32_t r = roots.back();
            roots.pop_back();
            cluster.clear();
            frontier.clear();
            cluster.push_back(r);
            if (n_interior + cluster.size() > nn) return fail(ctx, "rt_scene_upload: the node array is not a tree (cycle)");
            for (size_t head = 0; head < cluster.size(); ++head)       // BFS inside the treelet
            {
                uint32_t n = cluster[head];
                uint32_t kids[2] = {n + 1, sd->nodes[n].offset};
                for (uint32_t c : kids)
                {
                    if (c >= nn || c <= n) return fail(ctx, "rt_scene_upload: child index outside the node array");
                    if (!is_interior(c)) continue;
                    if (cluster.size() < kTreelet) cluster.push_back(c);
                    else frontier.push_back(c);
                }
            }
            for (uint32_t n : cluster) interior_index[n] = n_interior++;
            for (size_t k = frontier.size(); k-- > 0;) roots.push_back(frontier[k]);   // first child's cluster next
        }
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
        have_sh = choose_tree(sd, wide, w_entry, true, ctx->shadow_tree, own_sh, s.tree_report);
        if (have_sh) adopt_own(own_sh, s.wnodes_sh);
    }
    if (have_wide && n_wide_ref != 0u && may_own && ctx->closest_tree)
    {
        have_cl = choose_tree(sd, wide, w_entry, false, ctx->closest_tree, own_cl, s.tree_report);
        if (have_cl) adopt_own(own_cl, s.wnodes_cl);
    }
    const uint32_t w_entry_sh = own_sh.entry, w_entry_cl = own_cl.entry;
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
        a->bvh2.assign(sd->nodes, sd->nodes + nn);
        a->roots = std::move(wide_roots);
        if (have_sh) { a->bvh2_sh = std::move(own_sh.bvh2); a->roots_sh = std::move(own_sh.roots); }
        if (a->mode.load() & 16u)
        {
            a->tri9.resize((size_t)nt * 9);
            for (uint32_t i = 0; i < nt; ++i)
            {
                const rt_triangle& t = sd->triangles[i];
                const rt_float3 v[3] = {t.v1.position, t.v2.position, t.v3.position};
                for (int k = 0; k < 3; ++k) { a->tri9[(size_t)i * 9 + 3 * k] = v[k].x; a->tri9[(size_t)i * 9 + 3 * k + 1] = v[k].y; a->tri9[(size_t)i * 9 + 3 * k + 2] = v[k].z; }
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
    s.n_wide_sh = have_sh ? (uint32_t)own_sh.wide.size() : 0u;
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
            "(shadow tree: built in %.3f, folded in %.3f) + choosing by proxy rays and uploading %.3f + adaptation state %.3f (%u triangles, %u nodes, %u + %u wide records)\n",
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t_upload).count(), t_layout, t_device, t_fold, folded_on_device ? "on the device" : "on host threads", t_own_wait,
            own_sh.build_seconds, own_sh.fold_seconds, t_choose, t_rest, nt, nn, s.n_wide, s.n_wide_sh);
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
    if (ensure_slots(f, n_samples ? n_samples : 1u) != RT_OK) return RT_ERROR;
    if (reserved) *reserved = f->slots;
    return RT_OK;
}

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
    char line[400];
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
            "shadow %.2f -> %.2f (%s); %.2f s on a worker thread\n", a->adaptations, a->o.size(), a->sh_o.size(), a->cost[0][0], a->cost[0][1], a->ok ? "adopted" : "kept",
            a->cost[1][0], a->cost[1][1], a->ok_sh ? "adopted" : "kept", a->seconds);
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
        const uint64_t truncated = g_truncated_walks.exchange(0);
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
    if (ensure_slots(f, n_samples < cap ? n_samples : cap) != RT_OK) return RT_ERROR;
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
