/* rt_hip.h -- C-ABI of the MI355X (gfx950) wavefront path-tracing backend.
 *
 * This library is the drop-in replacement for the reference's OpenCL wrapper
 * layer src/gpu_wrappers/cl_context.{hpp,cpp} (CLContext / CLKernel) together
 * with the device half of src/integrator/cl_pt_integrator.{hpp,cpp} (the
 * buffers it owns and the 11 kernels it launches).  A reference-side
 * `HIPPathTraceIntegrator : Integrator` binds these entry points one to one
 * (this repo ships that class in raytracing_amd/host/).
 *
 * Conventions: every function returns 0 on success, non-zero on failure; the
 * message is available from rt_last_error() (replaces the thrown CLException,
 * src/utils/cl_exception.hpp:109-123 -- the C++ host shim rethrows).  Plain
 * pointers and sizes only; records are the PODs of rt_types.h.  One HIP stream
 * per context with in-order semantics, like the reference's single in-order
 * command queue (cl_context.cpp:89).  No call synchronises with the host
 * except rt_finish, rt_buffer_read, rt_frame_read_* and rt_frame_get_stats.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stddef.h>
#include <stdint.h>
#include "rt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_ERROR 1
#define RT_MAX_BOUNCES_LIMIT 62u

typedef struct rt_ctx rt_ctx;        /* CLContext, cl_context.hpp:37-65 */
typedef struct rt_buffer rt_buffer;  /* cl::Buffer */
typedef struct rt_frame rt_frame;    /* per-integrator device state, cl_pt_integrator.hpp:80-120 */

/* ---- context: CLContext::CLContext / Finish (cl_context.cpp:47-94, hpp:49) */
int rt_ctx_create(int device_ordinal, rt_ctx** out);
int rt_ctx_destroy(rt_ctx* ctx);
int rt_finish(rt_ctx* ctx);
/* last error message of `ctx`, or of the calling thread when ctx == NULL */
const char* rt_last_error(rt_ctx* ctx);
/* device name / CU count (the device info dump of cl_context.cpp:70-83) */
int rt_ctx_device_info(rt_ctx* ctx, char* name, size_t name_len, int* compute_units, size_t* hbm_bytes);
/* the hipStream_t of this context (for interop with other HIP libraries) */
void* rt_ctx_stream(rt_ctx* ctx);
/* Page-lock a caller-owned host buffer (and release it): read-backs into it -- rt_frame_resolve every frame in the
 * reference's call pattern (cl_pt_integrator.cpp:677-684 has the GL image for that; headless there is only the host) --
 * then run at the PCIe rate instead of through the runtime's staging copies.  Optional: every entry point works with
 * pageable memory. */
int rt_host_register(rt_ctx* ctx, void* host_ptr, size_t bytes);
int rt_host_unregister(rt_ctx* ctx, void* host_ptr);
/* context options, effective at the next rt_scene_upload */
enum rt_ctx_option
{
    RT_CTX_OPT_TREELET_NODES = 0   /* BVH record layout: interior nodes per contiguous breadth-first cluster
                                      (default 7; 1 = the reference's depth-first order).  Layout only. */
    , RT_CTX_OPT_WIDE_BVH = 1      /* 1 (default): also build the 4-wide quantized tree of k_trace_w4, each 64-byte record
                                      folding the SAH-optimal frontier of up to four BVH2 nodes; 2: two BVH2 levels per
                                      record (round 2's rule); 0: BVH2 records only */
    , RT_CTX_OPT_SHADOW_TREE = 2   /* 1 (default): shadow rays walk a 4-wide tree of the backend's own where that is cheaper --
                                      built over the reference's LEAVES (src/bvh.cpp:67-221 fixes only those for an any-hit
                                      query) by a full-sweep SAH on the projected area along the scene's directional lights
                                      (+ 50 % isotropic; surface area when there are only point lights); rt_scene_upload
                                      walks it and the reference's topology with proxy shadow rays and keeps the own tree if it
                                      saves more than 10 % of the steps (rt_scene_tree_report).  Verdicts equal TraceBvh
                                      -DSHADOW_RAYS bit for bit on either (trace_bvh.cl:107-109,164-167).  2: the own tree
                                      unconditionally; 3: the own tree with the surface-area metric (A/B); 0: shadow rays share
                                      the closest-hit tree */
    , RT_CTX_OPT_CLOSEST_TREE = 3  /* 0 (default): closest-hit rays walk the reference's topology in the reference's order
                                      (bit-identical radiance).  1 (measured like the shadow tree) / 2 (unconditionally):
                                      TOLERANCE MODE -- they walk an own surface-area tree, near child first on ITS split axes:
                                      the hit differs from TraceBvh's where two candidate hits tie within the rounding of
                                      RayTriangle (trace_bvh.cl:157-162); validated by rel-L2 < 1e-4 and a differing-pixel
                                      count against oracle/_ref, never the default */
    , RT_CTX_OPT_ADAPTIVE_FOLD = 4 /* The 4-wide trees start with the fold rt_scene_upload makes (optimal for the surface-area visit
                                      probability).  Default 25 = bits 0 + 3 + 4.  Bit 0: the first rt_integrate of an uploaded scene
                                      traces a small probe frame through the stage API (same camera, 1/k of the resolution, ~32 K paths),
                                      the host counts how often those rays pass each box of the binary tree, and a worker thread folds both
                                      4-wide trees again to be optimal for THOSE frequencies; the new records replace the old ones
                                      between two rt_integrate calls once they are ready, and again when a frame's camera has
                                      left the view they were made for (3 % of the scene's diagonal, 20 degrees, a tenth of the
                                      field of view) -- at most once per RT_CTX_OPT_ADAPT_MIN_INTERVAL_MS.  EXACT: a fold decides
                                      which interior boxes are tested, never a hit or a verdict.  Bit 1:
                                      rt_integrate waits for the new fold (reproducible timing: bench.py, tests).  Bit 2: also for
                                      trees of fewer than 8192 nodes (tests).  Bit 3: the shadow rays' BINARY tree is first rotated for
                                      the probe rays' measured crossings (tree_rotate.h) -- any tree over the reference's leaves gives
                                      an any-hit query the reference's verdict.  Bit 4: the slots of every shadow record are stored
                                      likeliest occluder first -- k_trace_w4<shadow> looks at them in stored order, an any-hit
                                      verdict is an OR, an occluded ray stops at its first hit; the host finds the probe rays' nearest
                                      occluders itself and keeps the triangles' positions for that (36 bytes each).  Measured on the
                                      device in round 5 (profiles/r05_call01_*): bits 3 + 4 take the shadow trace of the headline scene
                                      from 0.314 to 0.258 ms per sample (6711 -> 6917 Mrays/s), every config's frame bit-identical.
                                      0: off.
                                      Takes effect at the next rt_scene_upload; rt_scene_tree_report carries the latest
                                      adaptation's line.  Costs: a host copy of the binary tree(s), 48 bytes per node, for as long
                                      as the scene lives; per adaptation a probe frame's launches on the context's stream (nothing
                                      waits for them: the queues come back through pinned memory, the worker uploads the new
                                      records itself and rt_integrate only exchanges pointers); the worker gives up within
                                      milliseconds when the scene is uploaded again. */
    , RT_CTX_OPT_ADAPT_MIN_INTERVAL_MS = 5 /* default 500: a camera that keeps leaving the adapted view (an orbit) starts at most one
                                      fold adaptation per this many milliseconds (bit 1 of RT_CTX_OPT_ADAPTIVE_FOLD -- wait for every
                                      adaptation: tests, bench.py -- is not rate-limited).  Takes effect at once. */
    , RT_CTX_OPT_DEVICE_FOLD = 7   /* 1 (default): the collapse of a binary tree into the 4-wide records of k_trace_w4 -- the dynamic programme over node x slots,
                                      the record roots, the slots' placement, the quantised boxes -- runs on the DEVICE (raytracing_amd/csrc/fold_kernels.h:
                                      five kernels; the reference's tree at rt_scene_upload, the shadow rays' own tree, and every re-fold of an
                                      adaptation, whose crossing counts are a device kernel too); 0: on host threads (build_wide_bvh, the same algorithm:
                                      the two are compared record for record in tests/test_gpu_device_fold.py, and the host's is the fallback when the device
                                      path fails).  Results do not depend on it.  Takes effect at the next rt_scene_upload. */
    , RT_CTX_OPT_TREE_BUILDER = 9  /* who builds the shadow rays' own binary tree (RT_CTX_OPT_SHADOW_TREE).  0: own_bvh.h's full-sweep SAH on host threads (rounds 4 - 5).  1: the DEVICE --
                                      PLOC (parallel locally-ordered clustering, Meister & Bittner 2017) over the reference's leaves in Morton order, the tree's own metric (projected
                                      area along the directional lights + an isotropic share) as the merge cost, then the same fold where the tree lies (raytracing_amd/csrc/ploc_kernels.h):
                                      0.38 -> 0.09 s for 2.45 M leaves, 1.7 -> 0.26 s for 8.7 M.  2 (default): both start; the device's candidate is ready first and is measured first
                                      (proxy rays, rt_scene_tree_report); if it wins that measurement the host's build is abandoned, otherwise the host's candidate is waited for and
                                      measured as before -- never a worse tree than rounds 4 - 5 chose, and the upload of the headline scene 0.54 -> 0.24 s.  Any binary tree over the
                                      reference's leaves gives an any-hit query the reference's verdict.  Takes effect at the next rt_scene_upload. */
    , RT_CTX_OPT_WIDE_LAYOUT = 8   /* 0: the 4-wide records in the fold's own (depth-first) order; 1: in PAIRS -- every record with interior slots at an even index,
                                      the child it hands most rays on to right behind it, i.e. in the same 128-byte line (the L2 of gfx950 fetches whole lines:
                                      a 64-byte record that misses pays for its line-mate anyway).  A permutation of the records: no result depends on it.
                                      Takes effect at the next rt_scene_upload. */
    , RT_CTX_OPT_ADAPT_WAIT = 6    /* 1 / 0: sets / clears bit 1 of RT_CTX_OPT_ADAPTIVE_FOLD (rt_integrate waits for an adaptation it has
                                      started) for the scene IN PLACE, at once; the context's option, which the next upload reads, stays
                                      (bench.py: the headline waits for its fold, the moving-camera leg runs as the library ships) */
};
int rt_ctx_set_option(rt_ctx* ctx, int option, uint32_t value);
/* The blue-noise sampler tables (src/utils/blue_noise_sampler.hpp: sobol_256spp_256d[256*256],
 * scramblingTile[128*128*8], rankingTile[128*128*8], values 0..255) that CLPathTraceIntegrator
 * uploads in its ctor (cl_pt_integrator.cpp:222-235).  Required before RT_OPT_SAMPLER = 1. */
int rt_upload_blue_noise_tables(rt_ctx* ctx, const int* sobol_256spp_256d, const int* scramblingTile,
    const int* rankingTile);

/* ---- buffers: cl::Buffer(ctx, flags, size, host_ptr) (cl_pt_integrator.cpp:178-186)
 *      WriteBuffer / ReadBuffer / CopyBuffer (cl_context.cpp:96-113).
 *      rt_buffer_read is BLOCKING (the reference's ReadBuffer is non-blocking
 *      and never waited on, a latent race this ABI does not reproduce). */
int rt_buffer_create(rt_ctx* ctx, size_t bytes, const void* init_or_null, rt_buffer** out);
int rt_buffer_destroy(rt_buffer* buf);
int rt_buffer_write(rt_buffer* buf, size_t offset, const void* src, size_t bytes);
int rt_buffer_read(rt_buffer* buf, size_t offset, void* dst, size_t bytes);
int rt_buffer_copy(rt_buffer* src, rt_buffer* dst, size_t src_offset, size_t dst_offset, size_t bytes);
void* rt_buffer_device_ptr(rt_buffer* buf);
size_t rt_buffer_size(rt_buffer* buf);

/* ---- scene: CLPathTraceIntegrator::UploadGPUData (cl_pt_integrator.cpp:373-456)
 * Host arrays in the reference's layouts; the device re-layout (child-pair BVH
 * nodes, pre-differenced trace triangles, 128-byte shading records) happens
 * behind this call.  Arrays are copied; the caller keeps ownership. */
typedef struct rt_scene_desc
{
    const rt_triangle* triangles;         uint32_t num_triangles;   /* Scene::GetTriangles(), BVH order */
    const rt_bvh_node* nodes;             uint32_t num_nodes;       /* AccelerationStructure::GetNodes() */
    const rt_packed_material* materials;  uint32_t num_materials;
    const rt_texture* textures;           uint32_t num_textures;
    const uint32_t* texture_data;         uint32_t num_texture_data;
    const rt_light* lights;               uint32_t num_lights;
    const uint32_t* emissive_indices;     uint32_t num_emissive;    /* uploaded, unused (hit_surface.cl:39) */
    const float* env_rgba;                uint32_t env_width, env_height;  /* Scene::GetEnvImage(), float RGBA */
    /* ---- opt-in extensions.  NULL / 0 = the reference's behaviour, bit for bit. ---- */
    const uint16_t* material_texture_indices;   /* 6 per material -- diffuse, specular, roughness, metalness, emission,
                                                   transparency; 0xFFFF = none.  When given, these replace the 8-bit texture
                                                   indices packed into rt_packed_material and with them the 255-texture limit
                                                   (INVALID_TEXTURE_IDX 0xFF, constants.h:35; PackAlbedo's assert, scene.cpp:55) */
    uint32_t flags;                             /* RT_SCENE_* */
} rt_scene_desc;
/* rt_scene_desc::flags */
#define RT_SCENE_EMISSIVE_NEE 1u   /* next-event estimation also samples the emissive triangles of emissive_indices (which the
                                      reference collects, scene.cpp:324-339, and passes to a kernel that ignores them,
                                      hit_surface.cl:39).  Changes the estimator, not the expected image: DESIGN.md section 7b */

int rt_scene_upload(rt_ctx* ctx, const rt_scene_desc* scene);

/* One fold adaptation per process GROUP instead of one per rank (N ranks that tile one image hold the same scene and would each probe, rotate and fold for
 * identical records): the context's current 4-wide records -- the closest-hit rays' and the shadow rays' (n_shadow == 0: they share), as adapted so far -- to
 * host buffers of `capacity` records each (records NULL: size query; entries2 = {closest entry, shadow entry}), and into another context that has uploaded
 * the SAME scene (same triangle order): they replace its own, its adaptation is switched off, refs are range-checked.  The launcher's channel carries the
 * bytes in between (bench.py: torch.distributed).  Any fold of the same tree is exact: results do not change. */
int rt_scene_export_folds(rt_ctx* ctx, void* closest_records, void* shadow_records, uint32_t capacity, uint32_t* n_closest, uint32_t* n_shadow, uint32_t* entries2);
int rt_scene_import_folds(rt_ctx* ctx, const void* closest_records, uint32_t n_closest, uint32_t entry_closest, const void* shadow_records, uint32_t n_shadow, uint32_t entry_shadow);

/* ---- frame: the per-pixel state CLPathTraceIntegrator allocates in its ctor
 * (cl_pt_integrator.cpp:188-259).  A frame renders a TILE of the full image:
 * the rows whose band index (row / band_height) is congruent to tile_rank
 * modulo tile_count.  tile_count == 1 is the whole image.  Random numbers are
 * keyed by GLOBAL pixel coordinates, so any tiling yields identical pixels. */
typedef struct rt_frame_desc
{
    uint32_t width, height;      /* full image */
    uint32_t tile_rank;          /* 0 .. tile_count-1 */
    uint32_t tile_count;         /* >= 1 */
    uint32_t band_height;        /* rows per interleaved band (>= 1) */
} rt_frame_desc;

/* A frame borrows its context: destroy every frame before rt_ctx_destroy. */
int rt_frame_create(rt_ctx* ctx, const rt_frame_desc* desc, rt_frame** out);
int rt_frame_destroy(rt_frame* frame);
/* number of rows / pixels this tile owns, and the global row of local row r */
uint32_t rt_frame_local_rows(rt_frame* frame);
uint32_t rt_frame_global_row(rt_frame* frame, uint32_t local_row);

/* ---- integrator state (Integrator public API, integrator.hpp:52-62) */
enum rt_option
{
    RT_OPT_MAX_BOUNCES = 0,    /* Integrator::SetMaxBounces, default 3 (integrator.hpp:91) */
    RT_OPT_WHITE_FURNACE = 1,  /* Integrator::EnableWhiteFurnace (-D ENABLE_WHITE_FURNACE) */
    RT_OPT_SAMPLER = 2,        /* Integrator::SetSamplerType: 0 = kRandom, 1 = kBlueNoise (-D BLUE_NOISE_SAMPLER) */
    RT_OPT_AOV = 3,            /* Integrator::SetAOV: 0 shaded colour, 1 diffuse albedo, 2 depth, 3 normal, 4 motion vectors
                                  (resolve_radiance.cl:25-29); per-pixel, so it works on tiles */
    RT_OPT_DENOISER = 4,       /* Integrator::EnableDenoiser: 1 = temporal reprojection (denoiser.cl) in the frame itself
                                  (needs tile_count == 1: it reprojects across rows); 2 = the frame prepares the filter's
                                  inputs only and rt_group_denoise runs it on the gathered image (the mode for tiles) */
    RT_OPT_TRACE_DROP_LAST_BOUNCE_RAYS = 5, /* 1 (default): do not emit the never-traced rays of the last bounce */
    RT_OPT_PROFILE_KERNELS = 6, /* 1: bracket every kernel launch with HIP events on the context stream */
    RT_OPT_TRACE_VARIANT = 7    /* traversal kernel: 0 = k_trace_v1 (per-ray loop; tiny launches); 8 / 9 = k_trace2 (exact BVH2
                                   walk in separate wave-uniform node / triangle / refill loops) with a 10+12 / 12+12 entry
                                   LDS stack (closest + shadow); 10 (11: 16-entry LDS stack) = k_trace_w4 (4-wide quantized
                                   tree, exact leaf re-test; rays it cannot take -- non-finite 1/dir -- go to k_trace2);
                                   5 (default) = auto: k_trace_w4 wherever the tree qualifies.  Results are identical for
                                   every value.  (1..4, 6, 7 -- round 1's flat state-machine kernel -- and 12..15 -- stack-size
                                   sweeps and the direct-visit form, which now IS k_trace_w4 -- were removed in round 3.) */
    , RT_OPT_TRACE_WAVES_PER_CU = 8 /* persistent-grid size of the trace kernels in waves per CU (0 = as many as fit) */
    , RT_OPT_SAMPLES_IN_FLIGHT = 9  /* rt_integrate traces this many consecutive samples per pixel concurrently
                                       (1..1024, allocated at once; 0 = auto, the default: as many (<= 1024) as keep
                                       the per-path buffers under ~144 GB, allocated as batches ask for them).  Results are bit-identical for every value: contributions
                                       are logged per path and replayed in the reference's order. */
    , RT_OPT_TRACE_SELECT_FORM_BOX = 10 /* validation: 1 = every ray uses the reference's compare+select min/max in the
                                       slab test (trace_bvh.cl:85-97); by default only rays whose 1/dir has a
                                       non-finite component do (the only ones for which v_min/v_max_f32 could
                                       differ).  Results are identical for both values. */
    , RT_OPT_TRACE_PACKET_BOUNCES = 11 /* removed in round 3 (the packet kernel lost on every measured launch:
                                       profiles/r02_packet_kernel_on_coherent_bounces.log); only 0 is accepted */
    , RT_OPT_DEBUG_ALLOC_LIMIT = 13 /* test hook: per-path buffer allocations for more than this many samples in flight
                                       fail as if the device were out of memory (0 = off) */
    , RT_OPT_PATH_STATE_LIMIT_MB = 14 /* upper bound (MiB) for the per-path buffers (ray queues + radiance log, 412 B per
                                       path at 8 bounces): rt_integrate then runs every batch of samples chunk by chunk
                                       over the tile's pixels instead of over the whole tile at once.  0 (default) = only
                                       the built-in rule (at most half of the HBM).  Results are bit-identical for every
                                       value: path ids, the log and the replay are per pixel. */
    , RT_OPT_PIPELINES = 15        /* 1..4 (default 1): pipes -- sets of per-path buffers, each with its own HIP stream --
                                       rt_integrate deals the tile's chunks to when a batch is large (>= 2 samples in
                                       flight, >= 4 M paths), so that chunks overlap.  Measured: no gain on MI355X
                                       (profiles/r02_pipelines_sweep.log), hence off by default.  Results are bit-identical
                                       for every value. */
    , RT_OPT_SHADE_PARTITION = 16  /* bit 0: k_shade processes each block's 512 queue entries hits first, misses last, so that
                                       a wave runs either the surface code or the environment lookup, not both; bit 1: each
                                       block's outgoing and shadow rays enter their queues grouped by direction octant, so that
                                       a wave of the next trace launch holds rays that start near each other and point the
                                       same way (closest-hit trace -2.8 %).  Default 3.  Results are identical for every value. */
    , RT_OPT_OVERLAP_SHADOW = 17   /* 1 (default): inside rt_integrate the shadow trace of bounce b runs on a second,
                                       lower-priority stream beside the closest-hit trace and k_shade of bounce b + 1
                                       (both sides depend on k_shade(b) only; the shadow queue is double-buffered), so the
                                       ~0.8 ms in which a launch's last rays drain does not idle the machine.
                                       0: every launch on one stream.  Results are identical for both values. */
    , RT_OPT_SMALL_LAUNCH_PATHS = 18 /* trace launches of fewer rays than this run k_trace_w4 in CHUNK mode -- a wave takes 64
                                       consecutive rays, finishes all of them, takes the next 64; chunks are assigned statically,
                                       no refill of single lanes, no hand-out atomics -- decided inside the kernel from the live
                                       queue counter.  It is what makes the reference's own call pattern (one Integrate() per
                                       frame, one sample per pixel in flight) fast, and the late bounces of any batch.  Default
                                       3 000 000 -- 8 000 000 in the instance small batches launch (RT_OPT_TRACE_TAIL_PATHS), whose
                                       chunks refill their lanes; setting the option sets both; 0 = never.  Results are identical
                                       for every value. */
    , RT_OPT_COMPACT_LOG = 19      /* 1: rt_integrate batches of >= 8 samples in flight keep the radiance log COMPACT: six inline
                                       entries per path (a path of the benchmark scene logs 2.7 on average, 1 % more than six) +
                                       overflow blocks, bump-allocated one bounce ahead, for an eighth of the paths -- 290 instead
                                       of 412 bytes per path at 8 bounces, 314 instead of 604 at 16 -- at ~1.5 % of the throughput
                                       (k_shade's allocation step).  A batch that runs the pool dry (long-lived paths: a closed,
                                       lit room) is discarded and repeated in the full layout, which the frame then keeps
                                       (rt_stats.log_fallbacks).  0: always the full layout.  2 (default): compact exactly when the
                                       caller bounds the path state (RT_OPT_PATH_STATE_LIMIT_MB != 0: larger chunks, +2.9 % at
                                       32 GiB), full otherwise.  The stage calls (rt_generate_rays ... rt_advance_sample) always run
                                       on the full layout -- they have no batch to repeat -- and re-allocate a compact frame once.
                                       Results are bit-identical for every value. */
    , RT_OPT_DEBUG_LOG_POOL_DIV = 20 /* test hook: the overflow pool holds paths / value blocks (default 8) */
    , RT_OPT_TRACE_TAIL_LANES = 21  /* k_trace_w4's loop D, in the instance that launches known to be small take (a batch of fewer
                                       than RT_OPT_SMALL_LAUNCH_PATHS paths: the reference's one-sample-per-frame pattern): when
                                       this many or fewer lanes of a wave are still busy and none of the others can be refilled,
                                       every busy lane fetches its next record -- wide node or triangle -- and takes its step
                                       in the SAME pass: one memory round trip per step of the chunk's last, longest rays
                                       instead of one per kind of lane.  Default 40 (sweep: 2228 / 2332 / 2479 / 2501 / 2488
                                       Mrays/s per frame at 0 / 16 / 32 / 40 / 48); 0 = off.  Results are identical for every value. */
    , RT_OPT_TRACE_TAIL_PATHS = 22  /* batches of fewer paths than this (tile pixels x samples in flight; default 100 000 000: it pays up to ~32 samples of a 1080p frame in flight and costs 2 - 3 % at 128) launch the
                                       k_trace_w4 instance that has loop D.  Results are identical for every value. */
    , RT_OPT_CHUNK_REFILL = 23      /* k_trace_w4's chunk mode (small launches): 1 (default) = a wave's statically assigned chunks are its
                                       private queue and a lane that finishes takes the next ray of it at once (no atomics, no
                                       machine-wide tail); 0 = round 3's form, a wave finishes all 64 rays of a chunk before it takes
                                       the next.  Results are identical for both. */
    , RT_OPT_STAGE_PIPES = 24       /* 1..4 (default 1): ONE sample per pixel in flight -- the stage API (the reference's frame-by-frame
                                      pattern, Render::RenderFrame -> Integrator::Integrate, src/render.cpp:197) and rt_integrate(f, 1) --
                                      is cut into this many chunks of the tile (>= 512 x 512 pixels), each travelling through the
                                      wavefront loop on a pipe (HIP stream + per-path buffers) of its own.  Every launch of that
                                      pattern is its own tail (a launch lasts as long as its longest ray); side by side the chunks'
                                      tails overlap.  Same image bit for bit (chunks are independent; path ids are chunk-relative).
                                      Not with RT_OPT_AOV / RT_OPT_DENOISER (whole tile); the debug readers want 1. */
    , RT_OPT_FRAME_KERNEL = 25      /* 0 (default) / 1 / k = 2 .. 64 (k chunks of 64 pixels per wave: more or fewer blocks than are resident; measured: never
                                      better than 1) / 255 (the choice is MEASURED: after four warm-up frames, frames 4 - 19 of a scene alternate between the stage kernels
                                      and k_frame, timed with HIP events around each frame's launches, and the faster way stays until the next rt_scene_upload:
                                      k_frame wins 1.4 - 1.8 x on scenes of up to ~1 M triangles and loses 7 - 10 % on 2.8 M / 10 M): ONE sample per pixel in flight through the stage API -- the reference's frame-by-frame pattern,
                                      Integrator::Integrate through the fifteen hooks -- as ONE launch: the stage calls of a sample are recorded while
                                      they come in the canonical order (rt_generate_rays; rt_intersect, rt_shade, rt_intersect_shadow for bounce 0 ..
                                      max_bounces; rt_advance_sample) and rt_advance_sample launches k_frame, in which every wave carries its own
                                      pixels through all the bounces (raytracing_amd/csrc/frame_kernels.h).  Any other order, a debug reader or the
                                      radiance between two stages replays the recorded stages with the stage kernels first.  Same radiance and ray
                                      counters bit for bit.  Not with AOVs / the denoiser, the compact log, RT_SCENE_EMISSIVE_NEE or profiling
                                      (those samples take the stage kernels). */
    , RT_OPT_SAMPLES_AHEAD = 26     /* 0 (default: off) / 1 (automatic depth: batches of ~16 M paths, i.e. 8 samples of a 1080p frame) / k = 2 .. 64 samples per batch;
                                      + 256: the two banks launch on a stream each (their batches overlap) instead of one after the other on one stream.
                                      The stage API -- the reference's frame-by-frame pattern, one Integrate() per frame at one sample per pixel,
                                      src/render.cpp:197 -- traces the NEXT samples of a standing camera ahead: after three samples without rt_reset the frame
                                      enqueues batches of 2, 4, .. k consecutive samples (rt_integrate's launches, its radiance log left unreplayed) into two
                                      banks beside its own stream, and a later sample that sits in a bank costs its Integrate() one replay of that sample's log
                                      slot -- the radiance after EVERY call is the reference's, bit for bit, sample by sample.  rt_reset, another camera or
                                      option, rt_scene_upload, rt_integrate and anything that looks between two stages drop what was traced ahead (at most
                                      2 k samples of device time, once; a camera that moves every frame never starts the mode).  The image of a frame is
                                      what it always was; what changes is WHEN the work is done: a launch of one sample per pixel is its own tail (3.3 ms per
                                      1080p frame of the 2.8 M-triangle scene where the rays are worth 1.6), a launch of k is not.  Costs: two more sets of
                                      per-path buffers for k samples in flight (within 64 GiB, or RT_OPT_PATH_STATE_LIMIT_MB); rt_stats' ray totals run ahead of
                                      its sample count by rt_stats.samples_ahead.  Not with AOVs / the denoiser / RT_OPT_STAGE_PIPES / profiling. */
    , RT_OPT_TRACE_TUNE = 12       /* k_trace2 (variants 8, 9) loop thresholds: value & 255 = lanes that must hold an
                                       interior node for a wave to stay in the node loop, value >> 8 & 255 = lanes that
                                       must wait at a triangle for another pass of the triangle loop, value >> 16 & 255 = rays a wave takes
                                       from the queue per hand-out / 16 (k_trace_w4; 7 bits), value >> 24 = the fewest rays per lane a wave of
                                       the persistent grid is started for (k_trace_w4: the grid follows the live queue counter, the
                                       reference's "@TODO: use indirect dispatch"; 255 = every wave), bit 23 = chunk mode for every
                                       launch (RT_OPT_SMALL_LAUNCH_PATHS).  0 in a field = its default.
                                       Results are identical for every value. */
};
int rt_set_option(rt_frame* frame, int option, uint32_t value);
int rt_set_camera(rt_frame* frame, const rt_camera* camera);       /* SetCameraData, cl_pt_integrator.cpp:365-371 */

/* ---- stages: the protected virtuals Integrator::Integrate() schedules
 * (integrator.hpp:65-79, integrator.cpp:27-59).  The HIP backend fuses
 *   Miss + ClearCounter x2 + HitSurface          -> rt_shade
 *   TraceBvh(SHADOW_RAYS) + AccumulateDirectSamples -> rt_intersect_shadow
 * so rt_shade_miss / rt_clear_* / rt_accumulate_direct are accepted no-ops
 * kept for schedule compatibility. */
int rt_reset(rt_frame* frame);                          /* Reset */
int rt_generate_rays(rt_frame* frame);                  /* GenerateRays */
int rt_intersect(rt_frame* frame, uint32_t bounce);     /* IntersectRays */
int rt_shade_miss(rt_frame* frame, uint32_t bounce);    /* ShadeMissedRays (fused into rt_shade) */
int rt_clear_outgoing_counter(rt_frame* frame, uint32_t bounce);   /* ClearOutgoingRayCounter (no-op) */
int rt_clear_shadow_counter(rt_frame* frame);           /* ClearShadowRayCounter (no-op) */
int rt_shade(rt_frame* frame, uint32_t bounce);         /* ShadeSurfaceHits (+ miss) */
int rt_intersect_shadow(rt_frame* frame, uint32_t bounce);  /* IntersectShadowRays (+ accumulate) */
int rt_accumulate_direct(rt_frame* frame);              /* AccumulateDirectSamples (fused, no-op) */
int rt_compute_aovs(rt_frame* frame);                   /* ComputeAOVs (after rt_intersect(frame, 0)); no-op unless an AOV or the denoiser is on */
int rt_advance_sample(rt_frame* frame);                 /* AdvanceSampleCount */
int rt_denoise(rt_frame* frame);                        /* Denoise (TemporalAccumulation); no-op unless RT_OPT_DENOISER */
int rt_copy_history(rt_frame* frame);                   /* CopyHistoryBuffers */
/* fast path: n_samples x Integrate() enqueued without returning to the caller */
int rt_integrate(rt_frame* frame, uint32_t n_samples);
/* The per-path buffers (ray queues + radiance log) are sized by the largest batch of samples
 * requested so far and grow inside rt_integrate when a larger one arrives.  This call sizes
 * them ahead of time for rt_integrate(n_samples) -- clamped to RT_OPT_SAMPLES_IN_FLIGHT --
 * and returns the samples the frame can now keep in flight (like vector::reserve; no
 * reference counterpart: the reference allocates per-pixel state once, cl_pt_integrator.cpp:204-257). */
int rt_frame_reserve_samples(rt_frame* frame, uint32_t n_samples, uint32_t* reserved);

/* ---- output.  ResolveRadiance (resolve_radiance.cl:31-86) headless: RGBA32F,
 * local_rows x width, row-major.  rt_frame_read_radiance returns the running
 * SUM over samples (radiance_buffer_), rt_frame_resolve the tonemapped image. */
int rt_frame_resolve(rt_frame* frame, float* host_rgba);
/* The same stage as the reference runs it every frame -- ResolveRadiance, then Finish() (cl_pt_integrator.cpp:677-684): when
 * this returns every kernel of the frame has run, and the tonemapped image is ON ITS WAY to host_rgba (a copy stream of its
 * own, double-buffered on the device: the reference resolves into a GL image and nothing crosses PCIe; headless the image
 * overlaps the next frame's tracing instead).  host_rgba holds the frame after rt_frame_present_wait (or the next
 * rt_frame_resolve / rt_frame_destroy); presenting again into the same buffer is fine (the copies are ordered). */
int rt_frame_present(rt_frame* frame, float* host_rgba);
int rt_frame_present_wait(rt_frame* frame);
int rt_frame_read_radiance(rt_frame* frame, float* host_rgba);
/* device pointer of the running-sum radiance (float4[local_rows*width]) for
 * device-side gathers (RCCL) without a host bounce */
void* rt_frame_radiance_device_ptr(rt_frame* frame);
uint32_t rt_frame_sample_count(rt_frame* frame);

/* ---- statistics: the queue counters the reference keeps in
 * ray_counter_buffer_[2] / shadow_ray_counter_buffer_ (cl_pt_integrator.hpp:85-86),
 * sampled per bounce and accumulated on the device. */
typedef struct rt_stats
{
    uint64_t closest_rays;        /* sum over samples and bounces of rays traced closest-hit */
    uint64_t shadow_rays;         /* ... of shadow rays traced */
    uint64_t samples;             /* Integrate() calls since the last reset */
    uint32_t last_active[64];     /* per-bounce counts of the most recent batch of samples in flight */
    uint32_t last_shadow[64];
    uint32_t samples_in_flight;        /* samples the per-path buffers hold at the moment */
    uint32_t samples_in_flight_limit;  /* != 0: a larger batch did not fit into device memory and was halved to this */
    uint64_t path_state_bytes;         /* size of the per-path buffers (ray queues + radiance log) */
    uint32_t stack_spills;             /* lane-steps of the traversal kernels with stack entries in the HBM spill area (beyond the
                                          LDS entries) since the last reset; a 32-bit diagnostic that wraps (the headline
                                          workload adds ~1.3 M per sample per pixel of the frame: read it over short runs) */
    uint32_t slow_rays;                /* rays with a non-finite 1/dir component that k_trace_w4 handed to the BVH2 kernel */
    uint32_t chunk_pixels;             /* pixels of the tile that travel through the wavefront loop together (the whole
                                          tile unless RT_OPT_PATH_STATE_LIMIT_MB splits it) */
    uint32_t pipelines;                /* pipes (HIP streams) the chunks are dealt to */
    uint32_t log_inline_entries;       /* != 0: the radiance log is in its compact layout with this many inline entries per path
                                          (RT_OPT_COMPACT_LOG); 0: the full layout, 2 (max_bounces + 1) entries per path */
    uint32_t log_fallbacks;            /* batches whose overflow pool ran dry and that were repeated in the full layout */
    uint32_t frame_kernel_samples;     /* samples of the stage API that went through k_frame in one launch (RT_OPT_FRAME_KERNEL), since the frame was created */
    uint32_t samples_ahead;            /* RT_OPT_SAMPLES_AHEAD: samples traced (or being traced) ahead of `samples` at the moment; closest_rays / shadow_rays
                                          INCLUDE their rays (the banks count per batch) */
    uint64_t samples_from_banks;       /* RT_OPT_SAMPLES_AHEAD: samples of the stage API that were replayed out of a batch traced ahead, since the frame was created */
} rt_stats;
int rt_frame_get_stats(rt_frame* frame, rt_stats* out);
/* RT_OPT_FRAME_KERNEL's per-wave rows of the latest k_frame launch (diagnostics: rays per bounce and 100 MHz ticks per phase of every wave;
 * raytracing_amd/csrc/frame_kernels.h).  out may be NULL (size query). */
int rt_frame_debug_frame_rows(rt_frame* frame, uint32_t* out, uint32_t capacity_rows, uint32_t* n_rows, uint32_t* row_words);

/* ---- per-kernel timing (RT_OPT_PROFILE_KERNELS): HIP-event durations of the
 * launches since the option was switched on / since the last call, summed per
 * kernel class.  Synchronises the stream. */
typedef struct rt_profile
{
    double ms_raygen, ms_trace_closest, ms_shade, ms_trace_shadow;
    uint32_t n_raygen, n_trace_closest, n_shade, n_trace_shadow;
} rt_profile;
int rt_frame_get_profile(rt_frame* frame, rt_profile* out);

/* D2D copy of the running-sum radiance (float4[local_rows*width]) into a caller
 * buffer on the same device (e.g. a tensor handed to an RCCL gather); ordered
 * on the context stream and completed on return. */
int rt_frame_copy_radiance(rt_frame* frame, void* device_dst);

/* ---- device groups: one image tiled over several GPUs, ONE collective.
 * The reference drives a single device (src/gpu_wrappers/cl_context.cpp:89: one queue on devices_[0]);
 * this is the multi-GPU extension the integrators' pixel independence allows: every rank renders the
 * interleaved row bands rt_frame_desc gives it (tile_rank / tile_count / band_height), nothing is exchanged
 * while rendering, and rt_group_gather_radiance is the one RCCL gather (ncclGather over xGMI) of the
 * accumulated radiance to the root, which also puts the bands back into image order.
 *   - rt_group_create: all ranks in THIS process (one host thread may drive them; ncclCommInitAll);
 *   - rt_group_unique_id + rt_group_join: one process per GPU (torch.distributed.run, mpirun): rank 0
 *     creates the id, the launcher's own channel carries its RT_GROUP_ID_BYTES to the others, all join.
 * RCCL is loaded on first use (dlopen librccl.so.1); groups are not needed for single-GPU work. */
typedef struct rt_group rt_group;
#define RT_GROUP_ID_BYTES 128
int rt_group_create(int n, const int* device_ordinals, rt_group** out);
/* rt_group_create without this library's own one-rank-per-device check, so that the list reaches ncclCommInitAll whatever it
 * holds and RCCL's own answer comes back through rt_group_last_error: the wiring test of the in-process path on a one-GPU box
 * ({0, 0} must fail with RCCL's refusal).  Not for products. */
int rt_group_create_unchecked(int n, const int* device_ordinals, rt_group** out);
int rt_group_unique_id(void* id_bytes, size_t capacity);
int rt_group_join(int nranks, int rank, const void* id_bytes, int device_ordinal, rt_group** out);
int rt_group_size(rt_group* group);                     /* ranks in the group */
int rt_group_local_count(rt_group* group);              /* ranks living in this process */
int rt_group_local_rank(rt_group* group, int i);        /* global rank of local member i */
/* What RCCL reports for the communicator of local member i: *comm_ranks = ncclCommCount, *comm_user_rank =
 * ncclCommUserRank (either may be NULL).  rt_group_size echoes the caller's argument; this is the library's own
 * count, the evidence that the gather really spans N ranks.  A local group (no RCCL) reports 0 / -1. */
int rt_group_comm_count(rt_group* group, int i, int* comm_ranks, int* comm_user_rank);
/* frames[i] = the frame of local member i (its tile_rank must be that member's rank, tile_count the group
 * size).  On the process that owns `root`: host_rgba (may be NULL) receives height x width RGBA32F running
 * sums in image order, *device_rgba (may be NULL) the device copy of the same (valid until the next gather).
 * Stream-ordered after the frames' pending work; returns when the image is complete. */
int rt_group_gather_radiance(rt_group* group, rt_frame* const* frames, int root, float* host_rgba, void** device_rgba);
/* Temporal denoiser across tiles (denoiser.cl:27-79 reprojects across rows): after every rank has rendered ONE sample
 * of its tile with RT_OPT_DENOISER = 2, one gather carries radiance + depth + motion vectors to the root, which runs
 * TemporalAccumulation against its own history, copies the history and resolves.  On the root's process:
 * host_resolved_rgba / host_radiance_rgba (each may be NULL) receive the tonemapped frame / the filtered radiance. */
int rt_group_denoise(rt_group* group, rt_frame* const* frames, int root, float* host_resolved_rgba, float* host_radiance_rgba);
/* All ranks on ONE device, device copies instead of RCCL (which refuses two ranks per GPU): the plumbing / test
 * transport for boxes with a single GPU.  Same calls, same results. */
int rt_group_create_local(int n, int device_ordinal, rt_group** out);
int rt_group_destroy(rt_group* group);
const char* rt_group_last_error(rt_group* group);

/* ---- debug / parity access: copy a ray queue back in the reference's layouts.
 * which: 0 = incoming queue of `bounce` (rays_buffer_[bounce&1]), 1 = shadow queue.
 * Returns the element count through *count; arrays may be NULL. */
int rt_frame_debug_read_queue(rt_frame* frame, int which, uint32_t bounce, rt_ray* rays, uint32_t* pixel_indices,
    rt_float4* payload /* throughput (which=0) or direct light sample (which=1) */,
    uint32_t capacity /* elements the arrays can hold; all arrays NULL = size query */, uint32_t* count);
int rt_frame_debug_read_hits(rt_frame* frame, rt_hit* hits, uint32_t count);

/* Launch timeline of the closest-hit wide-tree kernel (k_trace_w4), per bounce: when its first wave started, when
 * the first wave found the queue dry, when its last wave left, in ticks of the 100 MHz wall clock.  arm = 1 clears
 * the record and starts recording, arm = 0 stops and reads out[64][6] (0 where nothing ran): the three times, the
 * most traversal steps any ray took, the slowest ray's ticks from hand-out to retirement and its steps; then
 * out[384 + i] = waves (of all recorded launches) that left in the i-th 25 us after their launch's queue ran dry.
 * tools/launch_timeline.py */
int rt_frame_debug_timeline(rt_frame* frame, int arm, unsigned long long* out);

/* The 4-wide quantized tree rt_scene_upload builds for k_trace_w4 from the reference's LinearBVHNode[]
 * (host only, no device needed): 64-byte records {origin.xyz, meta, lo[3], hi[3], ref[4], order (64 bits)} --
 * see build_wide_bvh in rt_hip.hip.  collapse: 1 = the SAH-optimal frontier per record (what rt_scene_upload uses),
 * 2 = two BVH2 levels per record (RT_CTX_OPT_WIDE_BVH's values).  records may be NULL (count query); roots (optional, with
 * records) receives the index of the BVH2 node each record folds.  Fails when the tree does not qualify. */
int rt_debug_wide_bvh(const rt_bvh_node* nodes, uint32_t num_nodes, int collapse, void* records, uint32_t* roots, uint32_t capacity,
    uint32_t* num_records, uint32_t* entry_ref);

/* What the last rt_scene_upload measured when it chose the trees (one line per ray population; "" when it had no choice), then the
 * latest fold adaptation's line (RT_CTX_OPT_ADAPTIVE_FOLD).  The pointer is valid until the next rt_integrate or rt_scene_upload on
 * this context (an adaptation rewrites its line): copy it. */
const char* rt_scene_tree_report(rt_ctx* ctx);
/* The tree rt_scene_upload would give the shadow (shadow != 0) or closest-hit rays of this scene under RT_CTX_OPT_SHADOW_TREE /
 * RT_CTX_OPT_CLOSEST_TREE = mode (host only; needs sd->triangles, nodes, lights): its records and the report line. */
int rt_debug_choose_tree(const rt_scene_desc* sd, int shadow, uint32_t mode, void* records, uint32_t capacity, uint32_t* num_records,
    uint32_t* entry_ref, char* report, size_t report_len);
/* The backend's own binary tree over the LEAVES of a reference LinearBVHNode[] (own_bvh.h; host only): same linear layout,
 * leaves copied.  Metric of a box = iso_weight * (dx dy + dy dz + dz dx) / 2 + sum over dirs of the projected area along
 * that unit direction (3 floats each).  out_nodes may be NULL (count query: 2 * leaves - 1). */
int rt_debug_own_bvh(const rt_bvh_node* nodes, uint32_t num_nodes, double iso_weight, const float* dirs, uint32_t n_dirs,
    rt_bvh_node* out_nodes, uint32_t capacity, uint32_t* num_out);
/* rt_debug_wide_bvh with collapse = 1 and the SAH collapse weighing boxes by that metric (what rt_scene_upload does for
 * the own trees) */
int rt_debug_wide_bvh_metric(const rt_bvh_node* nodes, uint32_t num_nodes, double iso_weight, const float* dirs, uint32_t n_dirs,
    void* records, uint32_t capacity, uint32_t* num_records, uint32_t* entry_ref);

/* fold_kernels.h on its own: the SAH collapse of `nodes` run on ctx's device (metric: iso_weight < 0 = plain surface area, else as rt_debug_wide_bvh_metric;
 * weights: per node, or NULL) -- the records, the node each one tests, *seconds = what the device path took.  tests/test_gpu_device_fold.py compares it
 * with rt_debug_wide_bvh / rt_debug_wide_bvh_metric / rt_debug_adapt_fold record for record. */
int rt_debug_device_fold(rt_ctx* ctx, const rt_bvh_node* nodes, uint32_t num_nodes, double iso_weight, const float* dirs, uint32_t n_dirs, const double* weights,
    void* records, uint32_t* roots, uint32_t capacity, uint32_t* num_records, uint32_t* entry_ref, double* seconds);
/* RT_CTX_OPT_WIDE_LAYOUT = 1 on its own (host only): the records of a fold of `nodes` (and the node each one tests) permuted in place into (parent,
 * likeliest child) pairs, by the area of the children's boxes. */
int rt_debug_pair_layout(const rt_bvh_node* nodes, uint32_t num_nodes, void* records, uint32_t* roots, uint32_t num_records);
/* ploc_kernels.h on its own: a binary tree over the leaves of `nodes` built on ctx's device with the metric of rt_debug_own_bvh (the same layout comes back: out_nodes[2 leaves - 1];
 * NULL = count query); *seconds = the device path's time, *rounds = clustering rounds. */
int rt_debug_device_tree(rt_ctx* ctx, const rt_bvh_node* nodes, uint32_t num_nodes, double iso_weight, const float* dirs, uint32_t n_dirs, rt_bvh_node* out_nodes, uint32_t capacity,
    uint32_t* num_out, double* seconds, uint32_t* rounds, uint32_t radius /* 0 = the library's */, const float* frame_dir /* NULL = world axes */, double stretch);
/* ... and the host's fold for given per-node weights (what an adaptation folds with), for that comparison */
int rt_debug_wide_bvh_weights(const rt_bvh_node* nodes, uint32_t num_nodes, const double* weights, void* records, uint32_t* roots, uint32_t capacity,
    uint32_t* num_records, uint32_t* entry_ref);

/* RT_CTX_OPT_ADAPTIVE_FOLD's host half on its own (no device): the fold of `nodes` adapted to n_rays rays (origins_tmax: x, y, z, t_max per
 * ray; directions: x, y, z, - per ray) -- its records (and, optional, the node each one tests), and cost2 = {the surface-area fold's, the adapted
 * fold's} box passes at record roots per ray; *cheaper = the adapted fold would be adopted. */
int rt_debug_adapt_fold(const rt_bvh_node* nodes, uint32_t num_nodes, const float* origins_tmax, const float* directions, uint32_t n_rays,
    void* records, uint32_t* roots, uint32_t capacity, uint32_t* num_records, uint32_t* entry_ref, double* cost2, int* cheaper);

/* The shadow side of an adaptation exactly as the worker thread runs it (host only): `nodes` = the shadow rays' current binary tree under its
 * surface-area fold, `mode` = RT_CTX_OPT_ADAPTIVE_FOLD's value (bit 3: rotate the tree first, keep whichever fold is cheaper).  Out: the candidate's
 * records (+ the node each one tests), the tree they fold (out_tree[num_nodes]), cost2 = {current, candidate} box passes at record roots per ray,
 * *rotations.  Returns 1 = would be adopted, 0 = kept, < 0 = error (rt_last_error(NULL)). */
int rt_debug_adapt_shadow_side(const rt_bvh_node* nodes, uint32_t num_nodes, const float* origins_tmax, const float* directions, uint32_t n_rays, uint32_t mode,
    void* records, uint32_t* roots, uint32_t capacity, uint32_t* num_records, uint32_t* entry_ref, rt_bvh_node* out_tree, double* cost2, uint32_t* rotations,
    const rt_triangle* triangles /* for mode bit 4 (may be NULL otherwise) */, uint32_t num_triangles, uint32_t* reordered /* records whose slots moved */);

/* What rt_scene_upload / rt_ctx_destroy do to an adaptation in flight (host only): a worker is started on `nodes` and the rays given (used as both
 * populations) and abandoned after delay_ms.  Returns the milliseconds abandoning took (the worker gives up at its next check), < 0 on an error;
 * *had_finished = the worker was done already. */
double rt_debug_fold_abandon(const rt_bvh_node* nodes, uint32_t num_nodes, const float* origins_tmax, const float* directions, uint32_t n_rays, uint32_t mode,
    uint32_t delay_ms, int* had_finished);

/* RT_CTX_OPT_ADAPTIVE_FOLD bit 3's tree search on its own (host only; raytracing_amd/csrc/tree_rotate.h): the binary tree `nodes` (reference
 * layout) rotated to lower the number of box crossings of the rays given (as rt_debug_adapt_fold takes them) -- out_nodes[num_nodes] holds a binary
 * tree over the same leaves in the same layout; cost2 = crossings of interior boxes per ray before / after; *rotations = how many were made. */
int rt_debug_rotate_tree(const rt_bvh_node* nodes, uint32_t num_nodes, const float* origins_tmax, const float* directions, uint32_t n_rays, int max_passes,
    rt_bvh_node* out_nodes, double* cost2, uint32_t* rotations, int moves /* bit 0: child <-> grandchild, bit 1: grandchild <-> grandchild */,
    double min_gain /* a move must save more than this share of the crossings at its node */);

/* RT_CTX_OPT_ADAPTIVE_FOLD's trigger on its own (host only): 1 when camera `now` has left the view the folds were adapted to -- position by more
 * than 3 % of scene_diagonal, direction by more than 20 degrees, field of view by more than a tenth -- else 0; -1 on a NULL argument. */
int rt_debug_fold_view_left(const rt_camera* adapted, const rt_camera* now, double scene_diagonal);

/* ---- kernel self-test hooks (known-answer tests of the device math):
 * evaluates fn over n inputs on the device.  fn: 0 sin, 1 cos, 2 tan, 3 pow(a,b),
 * 4 atan2(a,b), 5 acos, 6 sqrt, 7 a/b, 8 SampleRandom(bits of a.. as uints) */
int rt_debug_eval(rt_ctx* ctx, int fn, const float* a, const float* b, float* out, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
