#!/usr/bin/env python
"""bench.py -- headline benchmark of the wavefront path-tracing hot path.

  python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: the wavefront loop (raygen,
then per bounce: closest-hit trace, miss+shade, shadow trace+accumulate, then the
replay of the radiance log) over the batch of samples the path keeps in flight --
`samples_per_step` samples per pixel of the full frame (128 for the default
config, fixed per config and independent of N, so K steps are the same work at
every N).  That is the unit rt_integrate() works in; the reference's Integrate()
is the same loop over one sample per pixel, and `ms_per_spp` is reported next to
`ms_per_step`.  Workload: the configuration BASELINE.json's metric is quoted on,
configs[3] "Amazon Lumberyard Bistro 1920x1080 256spp 8-bounce" -- it fits one
GPU, so N = 1 runs it whole and N > 1 tiles it.  The Bistro asset is a download
the reference does not ship (assets/download_bistro.bat), so by default the
deterministic stand-in of SURVEY.md section 8d is generated (raytracing_amd/
scenes.py: city_block, ~2.8 M triangles, 120 materials, textured); with
`--scene exterior.obj --flip-yz --scale 0.01` (run_bistro.bat:15) or a cache
written by `rt_render --save-cache` the real asset is rendered instead and the
line says `"data": "real"`.  Default K = 8 steps = 1024 spp, W = 1.  --config
2 / 3 / 5 select the other BASELINE configs' stand-ins.  Metric = BASELINE.json's:
Mrays/s, all bounces + shadow rays, counted by the device queue counters the
reference itself keeps (ray_counter_buffer_, shadow_ray_counter_buffer_).

N > 1: one rank per GPU.  Launched by the driver under torch.distributed.run
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment); `python bench.py
--gpus N` on its own re-executes itself under torch.distributed.run.  The frame is
split into interleaved 8-row bands, the scene is replicated, there is no
communication while rendering and ONE gather of the accumulated radiance to rank 0
inside the timed region -- rt_group_gather_radiance of the C-ABI, i.e. ncclGather
of RCCL over xGMI; torch.distributed (gloo) only carries the 128-byte group id,
the barriers and the timing reductions.  "scaling": "strong" (total work fixed).
--debug-shared-gpu puts all ranks on GPU 0 (RCCL refuses two ranks per device, so
the gather then goes through gloo): plumbing check on a one-GPU box.
--plumbing-only runs launch, rendezvous, gather and the JSON line without a GPU
(synthetic tiles, no rendering, "value": null).

Extra objects on the JSON line:
  per_frame    the reference's OWN call pattern beside the headline: K x Render::RenderFrame() =
               Integrator::Integrate() through the fifteen stage hooks, one sample per pixel per call,
               ResolveRadiance + host sync every frame (src/render.cpp:197): mrays_per_s, ms_per_frame --
               as HIPPathTraceIntegrator ships (RT_OPT_FRAME_KERNEL = 255: the backend's measured choice
               between its stage kernels and ONE k_frame launch per frame).  per_frame.frame_kernel: both ways
               forced, each with bit_identical_to_the_default_leg; per_frame.moving_camera: the camera turned
               every frame, with the library's asynchronous fold re-adaptation and with it switched off.
  adaptation / surface_area_fold   the headline runs on the fold adapted to its view, the adaptation outside the
               timer: what it took (seconds_to_adapted) and the same job on the fold rt_scene_upload makes.
  scaling_estimate   "measured": false -- tools/tile_efficiency.py's one-GPU timing of every rank's tile of an
               N-way split (profiles/r05_tile_efficiency.json); no node with more than one GPU was available.
  roofline     the dominant kernel (closest-hit traversal).  bound "hbm": `achieved` = HBM GB/s from the
               rocprofv3 --pmc passes of this workload (profiles/r05_trace_counters.json, made by
               tools/pmc_bench2.sh + tools/make_counters_json.py), `frac` = achieved / 8 TB/s, `traffic` =
               HBM bytes per launch, next to SURVEY 8d's algorithmic bytes; `units` = busy fractions of the
               vector ALU / scalar ALU / L1-texture-address path against calibrated ceilings;
               `latency_ceiling` = the kernel's useful traversal steps per second against the bare
               visit chain's (tools/visit_microbench.hip); `counters.stale` = the counter file was
               collected from another code object than the library now running.  Launch duration and
               rays per launch are measured live (HIP events on the library's streams).
  parity       at N = 1: the frame the cpu_baseline leg renders with the reference's own
               kernels is rendered again on the GPU (same samples, outside the timed
               region) and compared: bit_identical, rel_l2, non-finite pixels on both sides;
               rel_l2_vs_libm_build: the same frame rendered by the reference kernels over glibc's libm
               instead of the project's rt_detmath.h (an independent pin of the transcendentals);
               reference_self_rel_l2: those two builds of the REFERENCE against each other, no GPU involved
               (what its result moves by when only its builtin library changes); median / p99 per-pixel errors.
  cpu_baseline the reference's own OpenCL kernels compiled for x86-64 (oracle/_ref, kind
               "reference") or the C restatement (kind "port"), timed on this box's host
               cores on a bounded sample.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
LIGHT = ((-0.6, -1.5, 3.5), (15.0, 10.0, 5.0))   # reference main.cpp:58
COUNTERS_FILE = os.path.join(ROOT, "profiles", "r06_trace_counters.json")
CALIBRATION_FILE = os.path.join(ROOT, "profiles", "r06_fetch_size_calibration.json")


SURVEY_A_ACTIVE = [65536, 65536, 51957, 44567, 38383]      # CornellBox.obj 256x256, sample 0, max_bounces 4
SURVEY_A_SHADOW = [48811, 30720, 25639, 22474, 19484]
# BASELINE.json configs (index = position in "configs"); config 1 is the CPU plumbing case.
CONFIGS = {
    1: dict(width=256, height=256, bounces=4, samples_per_step=1, name="BASELINE configs[0] exactly: assets/CornellBox.obj as shipped, the reference's default "
            "camera and light (camera_controller.cpp:30-41, main.cpp:58), sampler kRandom; on the GPU here, with the reference's own "
            "kernels on the host cores as cpu_baseline (the config's own definition: SURVEY 8d 'Config 1')"),
    2: dict(width=1280, height=720, bounces=8, samples_per_step=256, name="BASELINE configs[1] stand-in: Cornell shell + %(blob)d-tri displaced "
            "blob (dragon mtl) + %(ball)d-tri sphere (teapot mtl)"),
    3: dict(width=1920, height=1080, bounces=3, samples_per_step=128, name="BASELINE configs[2] stand-in: ShaderBalls.mtl 3x3 material grid on "
            "tessellated spheres + floor + 3 emissive quads, loaded from a generated OBJ (GGX+Lambert heavy)"),
    4: dict(width=1920, height=1080, bounces=8, samples_per_step=128, name="BASELINE configs[3] stand-in: procedural 'city block' (boxes, props, "
            "displaced foliage, 120 materials, textured) ~2.8 M triangles in place of Bistro exterior"),
    5: dict(width=3840, height=2160, bounces=16, samples_per_step=16, name="BASELINE configs[4] stand-in: procedural dense foliage courtyard "
            "~10 M triangles in place of San Miguel (deep BVH, high divergence)"),
}


def finish_scene(scene):
    scene.add_directional_light(*LIGHT)
    scene.set_env_path(os.path.join(ROOT, "assets", "ibl", "CGSkies_0036_free.hdr"))
    return scene, scene.lib.rth_scene_num_triangles(scene.handle)


def build_scene(args, host, S, finish=True):
    if getattr(args, "scene", None):
        # a real asset: OBJ/MTL through the C++ loader, or a binary cache written by rt_render --save-cache
        scene = host.Scene(args.scene, scale=args.scale, flip_yz=args.flip_yz, wide_texture_indices=getattr(args, "wide_texture_indices", False))
    elif args.config == 1:
        scene = host.Scene(os.path.join(ROOT, "assets", "CornellBox.obj"))     # the one asset of BASELINE's configs the repository holds
    elif args.config == 3:
        import tempfile
        path = S.shader_balls_obj(tempfile.mkdtemp(prefix="rt_bench_"), 100_000)
        scene = host.Scene(path)
    elif args.config == 4:
        scene = host.Scene(arrays=S.city_block(2_800_000))
    elif args.config == 5:
        scene = host.Scene(arrays=S.dense_foliage(10_000_000))
    else:
        tris, mats = S.cornell_blob(args.blob_tris, args.ball_tris)
        scene = host.Scene(arrays=dict(triangles=tris, materials=mats))
    if not finish:
        return scene
    return finish_scene(scene)


def cpu_legs(args, scene_arrays, cam_small, small_w, small_h, cam_full):
    """(a) instrumented oracle pass -> nodes/tris per ray; (b) timed CPU baseline.  Returns
    (per_ray, baseline, reference image or None, samples in that image)."""
    from tests import _oracle, _ref
    orc = _oracle.Oracle(small_w, small_h, scene_arrays)
    orc.set_camera(cam_small)
    orc.set_max_bounces(args.bounces)
    t0 = time.time()
    # one sample, stage by stage (the schedule of Integrator::Integrate), so that the queues of every bounce can also be
    # walked by the CPU restatement of k_trace_w4 (oracle.c: orc_wide_trace): steps per ray of the production kernel
    wide_cnt = {False: np.zeros(10, np.uint64), True: np.zeros(10, np.uint64)}
    try:
        from tests.test_wide_bvh import wide_of
        from raytracing_amd import types as T
        wide, wide_entry = wide_of(scene_arrays["nodes"], getattr(args, "wide_collapse", 1))
    except Exception:
        wide = None
    n_small = small_w * small_h
    orc.stage("reset"); orc.stage("generate_rays")
    c = s = 0
    for bounce in range(args.bounces + 1):
        k = int(orc.buffer("ray_counter%d" % (bounce & 1), np.uint32, 1)[0])
        c += k
        if wide is not None:
            orc.wide_trace(wide, wide_entry, orc.buffer("rays%d" % (bounce & 1), T.ray, n_small)[:k], False, wide_cnt[False], direct=True)
        orc.stage("intersect", bounce)
        for stg, sargs in (("shade_miss", (bounce,)), ("clear_counters", (bounce,)), ("shade_hits", (bounce,))):
            orc.stage(stg, *sargs)
        ks = int(orc.buffer("shadow_ray_counter", np.uint32, 1)[0])
        s += ks
        if wide is not None:
            orc.wide_trace(wide, wide_entry, orc.buffer("shadow_rays", T.ray, n_small)[:ks], True, wide_cnt[True], direct=True)
        orc.stage("intersect_shadow")
        orc.stage("accumulate")
    orc.stage("advance")
    t_orc = time.time() - t0
    st = orc.stats()
    per_ray = dict(closest_nodes=st["closest_nodes"] / max(c, 1), closest_tris=st["closest_tris"] / max(c, 1),
                   shadow_nodes=st["shadow_nodes"] / max(s, 1), shadow_tris=st["shadow_tris"] / max(s, 1))
    if wide is not None:
        for key, cnt in (("closest", wide_cnt[False]), ("shadow", wide_cnt[True])):
            r = max(int(cnt[0]), 1)
            # a step = one pass of a lane through loop C (a wide node) or loop B (a leaf arrival whose exact box fails, or
            # one triangle): wide_visits + leaf_box_fails + triangle_tests (Oracle.WIDE_COUNTERS)
            per_ray[key + "_wide_visits"] = float(cnt[1]) / r
            per_ray[key + "_steps"] = float(cnt[1] + cnt[3] + cnt[4]) / r
    baseline, ref_img, ref_spp, libm_img = None, None, 0, None
    if not args.no_cpu_baseline and args.gpus == 1:          # the CPU baseline is timed at N = 1 only
        cores = os.cpu_count() or 1
        if _ref.available():
            # The reference appends to its ray queues with one same-address atomic per ray
            # (hit_surface.cl:138,173); on a many-core host that contention makes more threads
            # SLOWER, so the baseline first picks the best thread count on a small frame.
            best_t, best_v = 1, 0.0
            for t in sorted({c for c in (8, 16, 32, 64, cores) if c <= cores}):
                probe = _ref.RefIntegrator(640, 360, scene_arrays, threads=t)
                probe.set_camera(cam_small)
                probe.set_max_bounces(args.bounces)
                t0 = time.time()
                probe.integrate(1)
                v = sum(probe.ray_totals()) / (time.time() - t0)
                if v > best_v:
                    best_t, best_v = t, v
                del probe
            # timed leg: the SAME frame as the GPU run
            ri = _ref.RefIntegrator(args.width, args.height, scene_arrays, threads=best_t)
            ri.set_camera(cam_full)
            ri.set_max_bounces(args.bounces)
            ri.integrate(1)                                   # warm-up (page in, thread start); sample 0 of the image
            r0 = sum(ri.ray_totals())
            t0 = time.time()
            n = 0
            while True:
                ri.integrate(1)
                n += 1
                if time.time() - t0 >= args.cpu_seconds or n >= 64:
                    break
            dt = time.time() - t0
            rays = sum(ri.ray_totals()) - r0
            ref_img, ref_spp = ri.radiance()[..., :3].copy(), n + 1
            # the same frame, same samples, by the reference kernels over glibc's libm instead of rt_detmath.h: an
            # INDEPENDENT pin of the transcendentals all three sides otherwise share (OpenCL leaves their rounding
            # implementation-defined; the tolerance of the north star, 1e-4, is what such a difference may cost)
            libm_img = None
            if _ref.available(libm=True):
                del ri
                rl = _ref.RefIntegrator(args.width, args.height, scene_arrays, threads=best_t, libm=True)
                rl.set_camera(cam_full)
                rl.set_max_bounces(args.bounces)
                rl.integrate(ref_spp)
                libm_img = rl.radiance()[..., :3].copy()
                del rl
            baseline = dict(value=round(rays / dt / 1e6, 3), unit="Mrays/s", cores=best_t, kind="reference",
                            opencl_cpu_device=opencl_cpu_device(),
                            sample="%d spp of the same scene at %dx%d, %d bounces (%.1f s; the reference's unmodified "
                                   ".cl kernels compiled for x86-64, NDRange = parallel-for over %d threads -- the best of "
                                   "8/16/32/64/%d on this %d-CPU host; more threads are slower because of the reference's "
                                   "same-address queue atomics)"
                                   % (n, args.width, args.height, args.bounces, dt, best_t, cores, cores))
        else:
            baseline = dict(value=(c + s) / t_orc / 1e6, unit="Mrays/s", cores=1, kind="port",
                            sample="1 spp of the same scene at %dx%d, %d bounces (%.1f s, oracle/oracle.c, scalar)"
                                   % (small_w, small_h, args.bounces, t_orc))
    return per_ray, baseline, ref_img, ref_spp, libm_img


def opencl_cpu_device():
    """is there a real OpenCL CPU device on this box?  (clGetDeviceIDs(CL_DEVICE_TYPE_CPU) over
    every platform of the ICD loader.)  Returns its name, or None -- in which case the x86-64 build of the reference's
    unmodified .cl files (oracle/_ref) IS the 'OpenCL-on-CPU' baseline."""
    import ctypes as C
    for name in ("libOpenCL.so.1", "libOpenCL.so", "/opt/rocm/lib/libOpenCL.so.1"):
        try:
            cl = C.CDLL(name)
            break
        except OSError:
            cl = None
    if cl is None:
        return None
    try:
        n = C.c_uint(0)
        if cl.clGetPlatformIDs(0, None, C.byref(n)) != 0 or n.value == 0:
            return None
        plats = (C.c_void_p * n.value)()
        cl.clGetPlatformIDs(n.value, plats, None)
        CL_DEVICE_TYPE_CPU, CL_DEVICE_NAME = 1 << 1, 0x102B
        for p in plats:
            nd = C.c_uint(0)
            if cl.clGetDeviceIDs(C.c_void_p(p), C.c_ulong(CL_DEVICE_TYPE_CPU), 0, None, C.byref(nd)) != 0 or nd.value == 0:
                continue
            devs = (C.c_void_p * nd.value)()
            cl.clGetDeviceIDs(C.c_void_p(p), C.c_ulong(CL_DEVICE_TYPE_CPU), nd.value, devs, None)
            buf = C.create_string_buffer(256)
            cl.clGetDeviceInfo(C.c_void_p(devs[0]), CL_DEVICE_NAME, 256, buf, None)
            return buf.value.decode(errors="replace")
    except Exception:
        return None
    return None


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: run the same command under torch.distributed.run."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def plumbing_only(args, rank, world):
    """No GPU, no rendering: launch + rendezvous + per-rank tiles + the gather + the JSON line."""
    import torch
    import torch.distributed as dist
    from raytracing_amd import distributed as D
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h = args.width, args.height
    rows = D.tile_rows(h, rank, world, args.band_height)
    # a tile whose pixels say where they belong: (global row, column, rank)
    tile = torch.zeros((len(rows), w, 4), dtype=torch.float32)
    tile[..., 0] = torch.as_tensor(rows, dtype=torch.float32)[:, None]
    tile[..., 1] = torch.arange(w, dtype=torch.float32)[None, :]
    tile[..., 2] = float(rank)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    full = D.gather_image(tile, h, w, rank, world, args.band_height)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    if rank == 0:
        ok = bool((full[..., 0] == torch.arange(h, dtype=torch.float32)[:, None]).all()
                  and (full[..., 1] == torch.arange(w, dtype=torch.float32)[None, :]).all())
        owner = ((torch.arange(h) // args.band_height) % world).to(torch.float32)
        ok = ok and bool((full[..., 2] == owner[:, None]).all())
        print(json.dumps(dict(metric="Mrays/s (all bounces+shadow)", value=None, unit="Mrays/s", n_gpus=world, steps=args.steps,
                              warmup=args.warmup, higher_is_better=True, scaling="strong", plumbing_only=True,
                              gather=dict(transport="gloo (plumbing check)", ms=round(float(tmax.item()) * 1e3, 3), image_ok=ok,
                                          nranks=world),
                              config=dict(workload="no rendering: launch, rendezvous, tile gather and report only",
                                          width=w, height=h))), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def per_frame_leg(args, render, lib, frame, capi, frames, resolve=True, ahead=None):
    """The reference's own call pattern (src/render.cpp:172-204): K x Render::RenderFrame() = Integrator::Integrate()
    through the fifteen stage hooks, ONE sample per pixel per call, each frame ending with ResolveRadiance and its
    host synchronisation (cl_pt_integrator.cpp:677-684).  Untimed by the headline; reported beside it.
    ahead: RT_OPT_SAMPLES_AHEAD for this leg (None = as the frame has it).  With the mode on, the banks count rays per batch and
    run ahead of the sample count, so the rays of exactly the timed frames' samples are counted AFTERWARDS by tracing the same
    sample indices again with rt_integrate (deterministic; untimed) -- which also says whether the two sums are the same bits."""
    import numpy as np
    if ahead is not None:
        assert lib.rt_set_option(frame, capi.OPT_SAMPLES_AHEAD, ahead) == 0
    assert lib.rt_reset(frame) == 0
    render.set_resolve_every_frame(resolve)
    warm = 24
    for _ in range(warm):                                 # (HIPPathTraceIntegrator's default times both ways over a scene's first 20 frames: RT_OPT_FRAME_KERNEL = 255)
        render.render_frame()
    render.finish()
    st0 = render.stats()
    times = []
    t0 = time.perf_counter()
    for _ in range(frames):
        ta = time.perf_counter()
        render.render_frame()
        times.append(time.perf_counter() - ta)
    render.finish()
    dt = time.perf_counter() - t0
    st1 = render.stats()
    render.set_resolve_every_frame(False)
    rays = float((st1.closest_rays - st0.closest_rays) + (st1.shadow_rays - st0.shadow_rays))
    banked = int(st1.samples_from_banks - st0.samples_from_banks)
    out = dict(frames=frames, samples_in_flight=1, resolve_every_frame=bool(resolve),
               frames_through_k_frame=int(st1.frame_kernel_samples - st0.frame_kernel_samples))
    if banked or st1.samples_ahead:
        got = render.radiance().copy()
        assert lib.rt_set_option(frame, capi.OPT_SAMPLES_IN_FLIGHT, 8) == 0
        assert lib.rt_reset(frame) == 0
        render.render_samples(warm)
        render.finish()
        sa = render.stats()
        render.render_samples(frames)
        render.finish()
        sb = render.stats()
        rays = float((sb.closest_rays - sa.closest_rays) + (sb.shadow_rays - sa.shadow_rays))
        t = np.sort(np.asarray(times)) * 1e3
        out["samples_ahead"] = dict(frames_replayed_from_a_batch=banked, samples_ahead_at_the_end=int(st1.samples_ahead),
                                    ms_per_call_median=round(float(t[len(t) // 2]), 4), ms_per_call_p99=round(float(t[min(len(t) - 1, int(0.99 * len(t)))]), 4),
                                    ms_per_call_max=round(float(t[-1]), 4),
                                    bit_identical_to_rt_integrate_of_the_same_samples=bool(np.array_equal(render.radiance(), got, equal_nan=True)),
                                    rays="counted by an rt_integrate of the same %d sample indices after the timed frames (the banks count per batch and run ahead)" % frames,
                                    what="RT_OPT_SAMPLES_AHEAD (HIPPathTraceIntegrator's default): while the camera stands still the next samples are traced ahead in "
                                         "batches of 2, 4, .. k on streams beside the frame's, and an Integrate() whose sample is there replays that sample's log slot -- "
                                         "the radiance after every call is the same bit for bit; calls come in bursts (a batch's samples one after the other, then a wait "
                                         "for the next batch): median / p99 / max per call beside the mean")
        assert lib.rt_set_option(frame, capi.OPT_SAMPLES_IN_FLIGHT, args.samples_in_flight) == 0
        assert lib.rt_reset(frame) == 0
    out.update(mrays_per_s=round(rays / dt / 1e6, 1), ms_per_frame=round(dt * 1e3 / frames, 4), rays_per_frame=round(rays / frames, 1),
               call_pattern="K x Render::RenderFrame() -> Integrator::Integrate() through the 15 stage hooks of HIPPathTraceIntegrator, "
                            "1 sample per pixel per call, ResolveRadiance + Finish() (host sync on the frame's kernels) every frame "
                            "(src/render.cpp:197, src/integrator/integrator.cpp:27-59, cl_pt_integrator.cpp:677-684); the resolved image "
                            "travels to the host on a copy stream while the next frame is traced (rt_frame_present: the reference "
                            "resolves into a GL image and reads nothing back); the last image's arrival is inside the timed region")
    return out


def frame_kernel_legs(args, render, lib, frame, capi, default_leg):
    """`per_frame` runs as HIPPathTraceIntegrator ships: RT_OPT_FRAME_KERNEL = 255, the backend's measured choice between its stage kernels and ONE
    launch of k_frame per Integrate() (the first dozen frames of a scene time both).  Beside it: both ways forced, same frames -- and the image after
    the same number of frames must be the same bit for bit whichever way they went."""
    import numpy as np
    frames = args.per_frame_frames
    out = {}
    try:
        # every Integrate() traces its own sample (RT_OPT_SAMPLES_AHEAD = 0: round 5's default): the measured choice, then both ways forced
        one = per_frame_leg(args, render, lib, frame, capi, frames, ahead=0)
        want = render.radiance().copy()                      # 24 + frames frames since its reset
        out["one_sample_per_call"] = dict(mrays_per_s=one["mrays_per_s"], ms_per_frame=one["ms_per_frame"], frames=frames, frames_through_k_frame=one["frames_through_k_frame"])
        out["default_went"] = "through k_frame" if one["frames_through_k_frame"] >= frames else "through the stage kernels"
        for name, mode in (("stage_kernels", 0), ("k_frame", 1)):
            assert lib.rt_set_option(frame, capi.OPT_FRAME_KERNEL, mode) == 0
            leg = per_frame_leg(args, render, lib, frame, capi, frames, ahead=0)
            out[name] = dict(mrays_per_s=leg["mrays_per_s"], ms_per_frame=leg["ms_per_frame"], frames=frames,
                             frames_through_k_frame=leg["frames_through_k_frame"],
                             bit_identical_to_the_default_leg=bool(np.array_equal(render.radiance(), want, equal_nan=True)))
    except Exception as e:                                   # noqa: BLE001 -- reported, never fatal to the measurement
        out["error"] = repr(e)
    finally:
        lib.rt_set_option(frame, capi.OPT_FRAME_KERNEL, 255)
        lib.rt_set_option(frame, capi.OPT_SAMPLES_AHEAD, 1 if args.samples_ahead is None else args.samples_ahead)
    out["what"] = ("with RT_OPT_SAMPLES_AHEAD = 0 (every Integrate() traces its own sample).  RT_OPT_FRAME_KERNEL: every frame of the hooks' pattern as one launch in which each wave carries its own pixels through all the bounces "
                   "(raytracing_amd/csrc/frame_kernels.h), or the stage kernels (47 launches per frame); 255 = measured per scene")
    return out


def cold_job_leg(args, render, host, capi, lib, torch, setup_breakdown):
    """BASELINE's job as a user of rt_render gets it, and the FIRST thing this process does with the device: the scene upload Render's constructor has just made
    (library defaults: RT_CTX_OPT_ADAPTIVE_FOLD = 25, the probe, the worker and the adoption run beside the job and nothing waits for them) + `--cold-job-spp` samples
    + the copy of the frame to the host, no reservation (rt_integrate sizes the job's buffers itself -- lean growth -- and maps them inside render_s).
    First, because a process that has just given large buffers back waits for the driver's wipe of them in its next hipMalloc (tools/alloc_microbench.hip,
    profiles/r06/call13_alloc_sequences.log: 0 - 6 s for 100 GiB, by what was released when) -- which is this bench's history, not a cold job's."""
    frame = host.load().rth_render_frame_handle(render.handle)
    tile = torch.zeros((max(render.local_rows, 1), args.width, 4), dtype=torch.float32, device="cuda")      # (torch's own start-up: not the job's)
    torch.cuda.synchronize()
    t_r = time.perf_counter()
    render.set_camera(host.default_camera(args.width, args.height))
    render.set_max_bounces(args.bounces)
    render.set_resolve_every_frame(False)
    c0 = render.stats()
    render.render_samples(args.cold_job_spp)
    render.finish()
    lib.rt_frame_copy_radiance(frame, tile.data_ptr())
    img = tile[:render.local_rows].cpu()
    t_c = time.perf_counter()
    c1 = render.stats()
    rays = float((c1.closest_rays - c0.closest_rays) + (c1.shadow_rays - c0.shadow_rays))
    upload_s = float(setup_breakdown.get("upload", 0.0))
    out = dict(spp=args.cold_job_spp, upload_s=round(upload_s, 3), samples_in_flight=int(c1.samples_in_flight), path_state_GB=round(c1.path_state_bytes / 2.0**30, 2),
               render_s=round(t_c - t_r, 3), wall_s=round(upload_s + (t_c - t_r), 3), rays=rays,
               non_finite_pixels=int((~torch.isfinite(img[..., :3]).all(-1)).sum().item()),      # (the reference's own: material.h:79-81 inf * 0 -- config.non_finite_pixels)
               mrays_per_s_render=round(rays / (t_c - t_r) / 1e6, 1), mrays_per_s_wall=round(rays / (upload_s + (t_c - t_r)) / 1e6, 1),
               trees=render.tree_report().strip().split("\n"),
               what="the config's whole job, cold, the first use of the device by this process: Render's own UploadGPUData (rt_scene_upload: re-layout, folds, own tree, tree choice) "
                    "+ %d spp + the frame's copy to the host, library defaults (adaptive fold 25 = asynchronous: the job starts on the upload's fold and adopts the adapted one when "
                    "its worker is done; no reservation: the buffers grow lean, an eighth of the samples asked for, mapped inside render_s); scene generation / OBJ parsing and the "
                    "reference-topology BVH build are in setup_s" % args.cold_job_spp)
    assert lib.rt_reset(frame) == 0
    render.finish()
    del tile
    return out


def median_pixel_rel_err(a, b):
    """median over the pixels of |a - b| / max(|b|, 1e-6) (L2 over the channels): one firefly sample cannot move it, unlike rel-L2"""
    import numpy as np
    fin = np.isfinite(a).all(-1) & np.isfinite(b).all(-1)
    if not fin.any():
        return None
    d = np.linalg.norm(a[fin].astype(np.float64) - b[fin].astype(np.float64), axis=-1)
    n = np.maximum(np.linalg.norm(b[fin].astype(np.float64), axis=-1), 1e-6)
    return float(np.median(d / n))


def p99_pixel_rel_err(a, b):
    """... and the 99th percentile of the same per-pixel quantity"""
    import numpy as np
    fin = np.isfinite(a).all(-1) & np.isfinite(b).all(-1)
    if not fin.any():
        return None
    d = np.linalg.norm(a[fin].astype(np.float64) - b[fin].astype(np.float64), axis=-1)
    n = np.maximum(np.linalg.norm(b[fin].astype(np.float64), axis=-1), 1e-6)
    return float(np.percentile(d / n, 99.0))


def moving_camera_leg(args, render, lib, frame, capi, host, cam0, frames):
    """The reference's interactive pattern with a camera that MOVES (src/render.cpp:188-195: a changed camera requests a reset, every
    frame starts its accumulation again): the camera turns about the up axis by 0.66 degrees per frame, so it leaves the view the folds
    were adapted to (20 degrees) every ~30 frames.  The fold adaptation runs as the library ships it (asynchronous: probe enqueued behind
    the frame, worker thread, pointer exchange; at most one per RT_CTX_OPT_ADAPT_MIN_INTERVAL_MS).  Beside it: the SAME camera path with
    re-adaptation rate-limited away, so that the difference is what re-adapting costs (or brings) an orbiting camera."""
    import numpy as np, re
    ctx = host.load().rth_render_ctx_handle(render.handle)
    def turned(deg):
        c = cam0.copy()
        a = np.float32(np.deg2rad(deg))
        ca, sa = np.cos(a), np.sin(a)
        for vec in ("front", "up"):
            x, y = float(c[vec]["x"]), float(c[vec]["y"])
            c[vec]["x"], c[vec]["y"] = np.float32(ca * x - sa * y), np.float32(sa * x + ca * y)
        return c
    def probe_no():
        m = re.search(r"adaptive fold \(probe (\d+)\)", render.tree_report())
        return int(m.group(1)) if m else 0
    def run(min_interval_ms):
        assert lib.rt_ctx_set_option(ctx, 5, min_interval_ms) == 0
        assert lib.rt_reset(frame) == 0
        render.set_resolve_every_frame(True)
        for i in range(3):
            render.set_camera(turned(0.0)); render.render_frame()
        render.finish()
        n0 = probe_no()
        t0 = time.perf_counter()
        for i in range(frames):
            render.set_camera(turned(0.66 * (i + 1)))
            render.render_frame()
        render.finish()
        dt = time.perf_counter() - t0
        render.set_resolve_every_frame(False)
        return dt, probe_no() - n0
    assert lib.rt_ctx_set_option(ctx, 6, 0) == 0                       # RT_CTX_OPT_ADAPT_WAIT off: as the library ships
    t_fixed, n_fixed = run(0xFFFFFFFF)                                 # the folds stay the ones made for the start view
    t_moving, n_moving = run(500)                                      # the library's default
    render.set_camera(cam0)
    assert lib.rt_ctx_set_option(ctx, 6, 1 if args.adaptive_fold & 2 else 0) == 0
    return dict(ms_per_frame=round(t_moving * 1e3 / frames, 4), ms_per_frame_without_re_adaptation=round(t_fixed * 1e3 / frames, 4), frames=frames,
                degrees_per_frame=0.66, with_over_without=round(t_moving / t_fixed, 4),
                adaptations_adopted=n_moving, adaptations_adopted_without=n_fixed,
                what="one Integrate() per frame through the hooks, camera turned (-> reset) every frame, ResolveRadiance + Finish() every frame; "
                     "fold adaptation asynchronous (library default, at most one per 500 ms) against the same camera path with the folds of the start view kept")


def roofline_object(args, world, live_step, per_ray, isolated):
    """`roofline` for the dominant kernel, the closest-hit traversal (k_trace_w4<closest>).  Every number follows a stated
    formula from (a) what this run measured live with HIP events on the library's streams and (b) committed counter files
    under profiles/, each tied to the code object it was collected from (`stale` when that is not the library running now).

      bound / achieved / peak / frac   the north star's quantity: HBM traffic the counters saw per launch / launch duration,
                                       against the 8 TB/s peak.  frac = achieved / peak.
      traffic                          HBM bytes per launch = 128 x TCC_EA0_RDREQ_128B + 64 x .._64B + 32 x .._32B + WRITE_SIZE KiB (tools/make_counters_json.py;
                                       round 6: this rocprofv3's FETCH_SIZE tallies every read request at 64 bytes while the L2 fetches whole 128-byte lines --
                                       profiles/r06_fetch_size_calibration.json checks both against known byte counts per access pattern)
      hbm_all_kernels                  the same for every kernel of the path (k_shade is the HBM-leaning one), and the ceilings the calibration measured:
                                       streaming reads / writes, and random 64-byte records (whose every miss moves a 128-byte line)
      algorithmic                      SURVEY 8d's per-ray byte model x rays per launch, and traffic / algorithmic
      units                            busy fractions of the units the kernel can saturate, against calibrated ceilings
      (latency_ceiling, a model that the kernel exceeded by 15 %, is gone from the line: VERDICT r05, weak 11)"""
    bytes_closest = 48.0 + 32.0 * per_ray["closest_nodes"] + 36.0 * per_ray["closest_tris"]     # SURVEY 8d, per ray
    avg_ms = live_step["avg_launch_ms"] if live_step else 0.0
    rays_per_launch = live_step["rays_per_launch"] if live_step else 0.0
    live = dict(live_step or {}, kernel="closest-hit traversal (k_trace_w4<closest> + its k_trace2 follow-up)",
                note="HIP-event SPANS of one more, UNTIMED step in the timed region's own mode (the timed region itself carries no "
                     "instrumentation): the shadow trace of bounce b runs beside the closest-hit trace of bounce b + 1 on a second "
                     "stream, so these spans overlap and are NOT a cost breakdown (their sum exceeds ms_per_spp); live_isolated has the costs")
    # the kernel alone on the machine (one more, untimed step with every launch on one stream) is what the counters describe
    iso_ms = isolated["avg_launch_ms"] if isolated else avg_ms
    iso_rays = isolated["rays_per_launch"] if isolated else rays_per_launch
    achieved_grays = iso_rays / (iso_ms * 1e-3) / 1e9 if iso_ms > 0 else 0.0
    alg_bytes = iso_rays * bytes_closest
    out = dict(kernel="k_trace_w4<closest> (the dominant kernel: %.0f %% of the kernels' time)" %
               (100.0 * (isolated["kernel_ms_per_spp"]["trace_closest"] / max(sum(isolated["kernel_ms_per_spp"].values()), 1e-9)) if isolated else 0.0),
               bound="hbm", achieved=None, peak=HBM_PEAK_GBS, unit="GB/s", frac=None, traffic=None, live=live,
               algorithmic=dict(bytes_per_ray=round(bytes_closest, 1), nodes_per_ray=round(per_ray["closest_nodes"], 2),
                                tris_per_ray=round(per_ray["closest_tris"], 2), bytes_per_launch=round(alg_bytes, 0),
                                GBs=round(alg_bytes / (iso_ms * 1e-3) / 1e9, 1) if iso_ms > 0 else None,
                                ratio_to_hbm_peak=round(alg_bytes / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if iso_ms > 0 else None,
                                note="SURVEY 8d byte model (48 + 32 n_nodes + 36 n_tris per ray on the reference's BVH2): LOGICAL "
                                     "bytes of the reference's walk; the kernel visits a quarter as many (wide, 8-bit) nodes and "
                                     "serves them from L1 / L2, so this is a count, not traffic and not a ceiling"))
    from raytracing_amd import codeobj
    running = codeobj.code_object_sha256()
    try:
        doc = json.load(open(COUNTERS_FILE))
        k = doc["config_%d" % args.config]["closest"]
    except Exception:
        out["counters"] = dict(file=os.path.relpath(COUNTERS_FILE, ROOT), error="no counters for this config: achieved / frac / traffic unavailable")
        k = None
    if k is not None:
        # the counters speak for one code object AND one fold of the trees (the same kernel visits fewer records on an adapted fold)
        fold_now = "adapted to the frame's rays" if args.adaptive_fold else "surface area"
        stale = doc.get("_code_object_sha256") != running or doc.get("_fold", "surface area") != fold_now
        traffic = float(k["hbm_bytes_per_launch"])
        achieved = traffic / (float(k["avg_launch_ms"]) * 1e-3) / 1e9
        out.update(achieved=round(achieved, 1), frac=round(achieved / HBM_PEAK_GBS, 4), traffic=round(traffic, 0),
                   formula="achieved = traffic / avg_launch_ms of the profiled launches (%.3f ms; this run, alone on the machine: %.3f ms); "
                           "frac = achieved / peak; traffic = read requests by size (%s) + WRITE_SIZE KiB per launch" % (k["avg_launch_ms"], iso_ms, k.get("hbm_read_how", "FETCH_SIZE x 0.99")),
                   stale=bool(stale))
        # every kernel of the path against HBM, and against what the calibration kernels reach with the same counters
        try:
            cal = json.load(open(CALIBRATION_FILE))["patterns"]
            rand_line_TBs = cal["k_rand64"]["TBs"] * cal["k_rand64"]["sized_request_bytes_over_known"]      # useful 64-byte records/s x the 128-byte lines they move
            ceil = dict(stream_read_TBs=cal["k_read16"]["TBs"], stream_write_TBs=cal["k_write16"]["TBs"], random_64B_records_useful_TBs=cal["k_rand64"]["TBs"],
                        random_64B_records_line_traffic_TBs=round(rand_line_TBs, 3))
        except Exception:
            cal, ceil, rand_line_TBs = None, None, None
        allk = {}
        for name in ("closest", "shadow", "shade"):
            e = doc["config_%d" % args.config].get(name)
            if e:
                allk[name] = dict(avg_launch_ms=round(e["avg_launch_ms"], 3), read_GB=round(e.get("hbm_read_bytes_per_launch", 0.0) / 1e9, 3),
                                  write_GB=round(e.get("hbm_write_bytes_per_launch", 0.0) / 1e9, 3), GBs=round(e["hbm_GBs"], 1), frac_of_peak=round(e["hbm_frac"], 4),
                                  frac_of_random_line_ceiling=round(e["hbm_GBs"] / 1e3 / rand_line_TBs, 3) if rand_line_TBs and name != "shade" else None,
                                  frac_of_stream_read_ceiling=round(e["hbm_GBs"] / 1e3 / ceil["stream_read_TBs"], 3) if ceil and name == "shade" else None,
                                  l2_hit_rate=round(e["per_launch"]["l2_hit_rate"], 3), valu_busy=round(e["valu_busy"], 3), l1_ta_busy=round(e["l1_ta_busy"], 3))
        out["hbm_all_kernels"] = dict(kernels=allk, measured_ceilings=ceil, calibration=os.path.relpath(CALIBRATION_FILE, ROOT),
                                      note="the traversal kernels fetch 64-byte records: every L2 miss moves a 128-byte line, half of it unasked for -- their ceiling is the "
                                           "random-record kernel's line traffic, not the streaming rate; k_shade streams its queues and is priced against the streaming read rate")
        out["algorithmic"]["traffic_over_algorithmic"] = round(traffic / alg_bytes, 4) if alg_bytes > 0 else None
        out["units"] = dict(valu_busy=round(k["valu_busy"], 4), salu_busy=round(k["salu_busy"], 4), l1_ta_busy=round(k["l1_ta_busy"], 4),
                            hbm_frac=round(k["hbm_frac"], 4), valu_fast_opcode_fraction=k.get("valu_fast_opcode_fraction"),
                            l1_accesses_per_clock_per_cu=round(k["per_launch"]["l1_accesses"] / k["cycles_per_launch"] / 256.0, 3),
                            note="busy fraction of each unit's calibrated ceiling (tools/make_counters_json.py; 1.0 = saturated): "
                                 "the vector ALU and the L1 / texture-address path are both near theirs, HBM is not")
        out["counters"] = dict(file=os.path.relpath(COUNTERS_FILE, ROOT), per_launch=k.get("per_launch"), launches_profiled=k.get("launches_profiled"),
                               code_object_sha256=doc.get("_code_object_sha256"), running_code_object_sha256=running, fold=doc.get("_fold", "surface area"), running_fold=fold_now, stale=bool(stale),
                               how=doc.get("_how"))
    # Ceilings in the metric's own unit (Grays/s of this kernel alone on the machine), one per resource the counters or a
    # micro-benchmark can speak for; frac_of_ceiling = achieved / the lowest of them.
    ceilings = {}
    if k is not None and achieved_grays > 0:
        for name, busy in (("valu_issue", k["valu_busy"]), ("l1_texture_address", k["l1_ta_busy"]), ("hbm", k["hbm_frac"])):
            if busy and busy > 0:
                ceilings[name] = round(achieved_grays / float(busy), 3)
    if ceilings:
        lowest = min(ceilings, key=ceilings.get)
        out["ceilings"] = dict(grays=ceilings, binding=lowest, achieved_grays=round(achieved_grays, 3),
                               busiest_unit_fraction=round(achieved_grays / ceilings[lowest], 4),
                               formula="valu_issue / l1_texture_address / hbm = achieved_grays / that unit's busy fraction of its calibrated ceiling "
                                       "(units above: counters of the profiled launches of this code object); busiest_unit_fraction = achieved_grays / min(ceilings) "
                                       "= the largest busy fraction: a MODEL of headroom (the issue ceilings are self-calibrated, tools/make_counters_json.py), not a measurement of it")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8,
                    help="timed steps; one step = samples_per_step samples per pixel (see the module docstring)")
    ap.add_argument("--warmup", type=int, default=1, help="untimed warm-up steps")
    ap.add_argument("--samples-per-step", type=int, default=None, help="override the config's samples per step")
    ap.add_argument("--config", type=int, default=4, choices=sorted(CONFIGS),
                    help="BASELINE.json config (1-based index into 'configs'); default 4 = the one the metric is quoted on "
                         "(Bistro 1080p 8-bounce stand-in)")
    ap.add_argument("--scene", default=None, help="render this OBJ (with its MTL/textures) or .rtscene cache instead of the stand-in, "
                    "e.g. --scene exterior.obj --flip-yz --scale 0.01 for Bistro (run_bistro.bat:15)")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--flip-yz", action="store_true")
    ap.add_argument("--wide-texture-indices", action="store_true", help="with --scene: load more than 255 textures (opt-in extension)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--bounces", type=int, default=None)
    ap.add_argument("--blob-tris", type=int, default=871_200)
    ap.add_argument("--ball-tris", type=int, default=20_000)
    ap.add_argument("--band-height", type=int, default=8)
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--path-state-gb", type=float, default=0.0,
                    help="cap the per-path device buffers (ray queues + radiance log) at this many GiB per GPU: the tile is then "
                         "rendered chunk by chunk (RT_OPT_PATH_STATE_LIMIT_MB); 0 = the library's own rule (up to half of the HBM)")
    ap.add_argument("--pipelines", type=int, default=None, help="RT_OPT_PIPELINES (library default: 1)")
    ap.add_argument("--trace-tune", type=lambda x: int(x, 0), default=0, help="RT_OPT_TRACE_TUNE (0 = defaults)")
    ap.add_argument("--overlap-shadow", type=int, default=None, help="RT_OPT_OVERLAP_SHADOW (default: the library's, 1)")
    ap.add_argument("--samples-in-flight", type=int, default=0, help="RT_OPT_SAMPLES_IN_FLIGHT (0 = automatic)")
    ap.add_argument("--shade-partition", type=int, default=None, help="RT_OPT_SHADE_PARTITION (library default: 3)")
    ap.add_argument("--trace-waves", type=int, default=0, help="RT_OPT_TRACE_WAVES_PER_CU (0 = as many as fit)")
    ap.add_argument("--trace-variant", type=int, default=None, help="RT_OPT_TRACE_VARIANT (default: the library's automatic choice)")
    ap.add_argument("--small-launch-paths", type=int, default=None, help="RT_OPT_SMALL_LAUNCH_PATHS (library default 3000000)")
    ap.add_argument("--wide-collapse", type=int, default=1, choices=(1, 2),
                    help="RT_CTX_OPT_WIDE_BVH: 1 = SAH-optimal frontier per wide record (library default), 2 = two BVH2 levels per record (A/B)")
    ap.add_argument("--shadow-tree", type=int, default=None, help="RT_CTX_OPT_SHADOW_TREE (library default 1: the backend's own tree for shadow rays "
                    "where it measures cheaper; 2 always; 3 always, surface-area metric; 0 shared with the closest-hit rays).  Bit-identical for every value.")
    ap.add_argument("--closest-tree", type=int, default=None, help="RT_CTX_OPT_CLOSEST_TREE (library default 0 = bit-identical; 1 / 2 = TOLERANCE mode: "
                    "an own tree for closest-hit rays where it measures cheaper / always)")
    ap.add_argument("--adaptive-fold", type=int, default=27, help="RT_CTX_OPT_ADAPTIVE_FOLD (library default 25 = bits 0 + 3 + 4: the first integrate probes the "
                    "frame's own rays, a worker thread folds both 4-wide trees again for their measured box passes -- the shadow rays' binary tree rotated first, "
                    "their records' slots stored likeliest occluder first -- and the records are replaced when ready; 27 (here) = + bit 1: the warm-up "
                    "waits for the new fold, so that every timed step runs on it; 3 = round 4's default (fold only); 0 = the upload's surface-area fold).  "
                    "Bit-identical for every value.")
    ap.add_argument("--tail-lanes", type=int, default=None, help="RT_OPT_TRACE_TAIL_LANES (library default 40; 0 = loop D off)")
    ap.add_argument("--chunk-refill", type=int, default=None, help="RT_OPT_CHUNK_REFILL (library default 1)")
    ap.add_argument("--tail-paths", type=int, default=None, help="RT_OPT_TRACE_TAIL_PATHS (library default 100000000)")
    ap.add_argument("--libm-series", default=None, help="sample counts (e.g. 1,2,4,8) of parity.rel_l2_vs_libm_build_series: the HIP path against the "
                    "reference's kernels over glibc libm on a 960x540 frame of the same scene (default: 1,2,4,8 for --config 5, off elsewhere; '' = off)")
    ap.add_argument("--compact-log", type=int, default=None, help="RT_OPT_COMPACT_LOG (library default 2: compact only when the path state is bounded; 1 = always for batches of >= 8 samples; 0 = never)")
    ap.add_argument("--per-frame-frames", type=int, default=48, help="frames of the per_frame leg (the reference's call pattern, "
                    "one Integrate() per frame); 0 = skip it")
    ap.add_argument("--stage-pipes", type=int, default=None, help="RT_OPT_STAGE_PIPES for the per_frame legs (library default 1: the frame's one sample per pixel "
                    "travels as one chunk; 2..4: as that many chunks on streams of their own, their launch tails overlapping)")
    ap.add_argument("--frame-kernel", type=int, default=None, help="RT_OPT_FRAME_KERNEL for the per_frame legs (library default 0): 1 = every frame of the "
                    "hooks' pattern is ONE launch of k_frame")
    ap.add_argument("--share-folds", type=int, default=1, help="N > 1: 1 (default) = one fold adaptation per process group (rank 0's records are broadcast), 0 = every rank adapts itself")
    ap.add_argument("--wide-layout", type=int, default=None, help="RT_CTX_OPT_WIDE_LAYOUT (library default 0): 1 = the 4-wide records in (parent, likeliest child) pairs, one pair per 128-byte line")
    ap.add_argument("--tree-builder", type=int, default=None, help="RT_CTX_OPT_TREE_BUILDER (library default 0): 1 = the shadow rays' own binary tree built on the device (PLOC)")
    ap.add_argument("--device-fold", type=int, default=None, help="RT_CTX_OPT_DEVICE_FOLD (library default 1): 0 = the folds on host threads")
    ap.add_argument("--samples-ahead", type=int, default=None, help="RT_OPT_SAMPLES_AHEAD for the per_frame leg (HIPPathTraceIntegrator's default: 1 = automatic depth; "
                    "0 = every Integrate() traces its own sample; k = 2..64 samples per batch; + 256 = one stream per bank)")
    ap.add_argument("--moving-camera-frames", type=int, default=720, help="frames of per_frame.moving_camera (0 = skip): the camera turns 0.66 degrees per frame, "
                    "so it leaves the adapted view every ~30 frames, with the library's default (asynchronous) fold adaptation")
    ap.add_argument("--surface-area-fold-steps", type=int, default=2, help="steps of the surface-area-fold figure printed beside value (0 = skip): the scene uploaded "
                    "again with RT_CTX_OPT_ADAPTIVE_FOLD = 0 after everything else, untimed by the driver")
    ap.add_argument("--cold-job-spp", type=int, default=256, help="samples of the cold_job leg: BASELINE's job from rt_scene_upload to the gather on the library's defaults (0 = skip)")
    ap.add_argument("--per-frame-only", action="store_true", help="run only the per_frame leg and print its object (tuning runs)")
    ap.add_argument("--debug-shared-gpu", action="store_true",
                    help="plumbing test only: all ranks share GPU 0 and gather over gloo (RCCL refuses two ranks per device)")
    ap.add_argument("--debug-try-rccl", action="store_true", help="with --debug-shared-gpu: ask RCCL for the communicator anyway (it refuses two ranks "
                    "per device): exercises the agreement that falls back to the gloo gather")
    ap.add_argument("--plumbing-only", action="store_true", help="no GPU: launch, rendezvous, gather and report only")
    ap.add_argument("--no-scene-cache", action="store_true", help="N > 1: every rank builds the scene and the BVH itself")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    args.width = args.width or cfg["width"]
    args.height = args.height or cfg["height"]
    args.bounces = cfg["bounces"] if args.bounces is None else args.bounces

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    if args.plumbing_only:
        return plumbing_only(args, rank, world)

    import torch
    import torch.distributed as dist
    from raytracing_amd import capi, host, scenes as S, distributed as D

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if args.debug_shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    group, rccl_fallback = None, None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # control plane only (group id, barriers, timing reductions): gloo.  The data-path collective is RCCL
        # behind the C-ABI (rt_group_gather_radiance).
        dist.init_process_group("gloo", rank=rank, world_size=world)
        if not args.debug_shared_gpu or args.debug_try_rccl:
            # RCCL communicator over the C-ABI.  If it cannot be had on EVERY rank (library missing, init error), all ranks
            # agree to gather over gloo through host memory instead and the line says so: a run on a node this code has
            # never seen must not lose its measurement to the transport.
            try:
                ids = [capi.Group.unique_id() if rank == 0 else None]
            except Exception as e:                          # noqa: BLE001
                ids, rccl_fallback = [None], "rt_group_unique_id: %r" % (e,)
            dist.broadcast_object_list(ids, src=0)
            if ids[0] is not None:
                try:
                    group = capi.Group.join(world, rank, ids[0], local_rank)
                except Exception as e:                      # noqa: BLE001
                    rccl_fallback = "rt_group_join on rank %d: %r" % (rank, e)
            okay = torch.tensor([1 if group is not None else 0], dtype=torch.int32)
            dist.all_reduce(okay, op=dist.ReduceOp.MIN)
            if int(okay.item()) == 0:
                if group is not None:
                    group.close()
                    group = None
                reasons = [None] * world
                dist.all_gather_object(reasons, rccl_fallback)
                rccl_fallback = next((r for r in reasons if r), "another rank could not join")

    # ---- setup (untimed): scene, BVH, upload -------------------------------
    t0 = time.time()
    if world > 1 and not args.no_scene_cache:
        # ONE rank parses / generates the scene and builds the BVH; it leaves both in a binary scene cache
        # (Scene::SaveCache) the other ranks load -- N ranks on one host would otherwise do the same host work N times
        import tempfile
        cache = os.path.join(tempfile.gettempdir(), "rt_bench_%s_%s_cfg%d.rtscene" % (os.environ.get("MASTER_ADDR", "local").replace(":", "_"),
                                                                                   os.environ.get("MASTER_PORT", "0"), args.config))
        if rank == 0:
            raw = build_scene(args, host, S, finish=False)
            raw.save_cache(cache)
            raw.close()
        dist.barrier()
        scene, n_tris = finish_scene(host.Scene(cache))
        scene_source = "rank 0 built the scene + BVH once and wrote a scene cache; every rank loaded it"
    else:
        scene, n_tris = build_scene(args, host, S)
        scene_source = "built by this rank"
    t_scene = time.time() - t0
    # N > 1: ONE fold adaptation per process group: rank 0 builds the shadow rays' tree and adapts the folds, the other ranks upload without either
    # (RT_CTX_OPT_SHADOW_TREE = 0, RT_CTX_OPT_ADAPTIVE_FOLD = 0) and take rank 0's records after its warm-up (rt_scene_export_folds -> this launcher's
    # gloo broadcast -> rt_scene_import_folds); any fold is exact, so the image does not depend on it
    share_folds = world > 1 and args.share_folds and args.adaptive_fold != 0 and args.shadow_tree is None and args.closest_tree is None
    render = host.Render(args.width, args.height, scene, device=local_rank, tile_rank=rank, tile_count=world,
                         band_height=args.band_height,      # builds the BVH (or adopts the cached one), finalises, uploads
                         ctx_options=((2, 0), (4, 0)) if share_folds and rank != 0 else ())
    t_setup = time.time() - t0
    setup_breakdown = render.setup_seconds()                 # Render's constructor: BVH build / Finalize / frame / UploadGPUData (its stages: the `upload:` line of the tree report)
    cold_job = None
    if world == 1 and args.cold_job_spp > 0 and not args.per_frame_only:
        try:
            cold_job = cold_job_leg(args, render, host, capi, capi.load(), torch, setup_breakdown)
        except Exception as e:                                # noqa: BLE001 -- reported, never fatal to the measurement
            cold_job = dict(error=repr(e))
        if not (args.shadow_tree is not None or args.closest_tree is not None or args.adaptive_fold != capi.ADAPTIVE_FOLD_DEFAULT or args.wide_layout is not None
                or args.device_fold is not None or args.tree_builder is not None or args.wide_collapse != 1):
            render.set_wide_bvh(args.wide_collapse)           # the measured job starts from a fresh upload as well (no other branch below makes one)
    if args.wide_collapse != 1:
        render.set_wide_bvh(args.wide_collapse)               # A/B: uploads the scene again with the other collapse
    if share_folds and rank != 0:
        pass                                                  # (this rank's folds come from rank 0: nothing to set, nothing to upload again)
    elif args.shadow_tree is not None or args.closest_tree is not None or args.adaptive_fold != capi.ADAPTIVE_FOLD_DEFAULT or args.wide_layout is not None or args.device_fold is not None or args.tree_builder is not None:
        if args.tree_builder is not None:
            render.set_ctx_option(9, args.tree_builder, False)     # RT_CTX_OPT_TREE_BUILDER
        if args.wide_layout is not None:
            render.set_ctx_option(8, args.wide_layout, False)      # RT_CTX_OPT_WIDE_LAYOUT
        if args.device_fold is not None:
            render.set_ctx_option(7, args.device_fold, False)      # RT_CTX_OPT_DEVICE_FOLD
        if args.adaptive_fold != capi.ADAPTIVE_FOLD_DEFAULT:
            render.set_adaptive_fold(args.adaptive_fold, upload=False)
        if args.shadow_tree is not None:
            render.set_shadow_tree(args.shadow_tree, upload=False)
        if args.closest_tree is not None:
            render.set_closest_tree(args.closest_tree, upload=False)
        render.set_wide_bvh(args.wide_collapse)               # ... and uploads again
    tree_report = render.tree_report()
    if world > 1 and not args.no_scene_cache:
        dist.barrier()
        if rank == 0:
            try:
                os.remove(cache)
            except OSError:
                pass
    cam = host.default_camera(args.width, args.height)
    render.set_camera(cam)
    render.set_max_bounces(args.bounces)
    render.set_resolve_every_frame(False)
    frame = host.load().rth_render_frame_handle(render.handle)
    lib = capi.load()

    def sync():
        render.finish()
        torch.cuda.synchronize()

    local_rows = render.local_rows
    tile = torch.zeros((max(local_rows, 1), args.width, 4), dtype=torch.float32, device="cuda")

    def gather(want_host):
        """The one collective.  Returns the full image on rank 0 (torch CPU tensor) when want_host."""
        if group is not None:
            img = group.gather_radiance([frame], 0, args.height, args.width, want_host=want_host)
            return torch.from_numpy(img) if img is not None else None
        if local_rows:
            lib.rt_frame_copy_radiance(frame, tile.data_ptr())
        if world == 1:
            return tile[:local_rows].cpu() if want_host else None
        full = D.gather_image(tile[:local_rows].cpu(), args.height, args.width, rank, world, args.band_height)   # debug: gloo
        return full

    # per-path buffers sized for the K-sample job before anything is timed (they would
    # otherwise grow inside the first rt_integrate that asks for a larger batch)
    sps = args.samples_per_step or cfg["samples_per_step"]
    spp_timed, spp_warm = args.steps * sps, args.warmup * sps
    if args.pipelines:
        assert lib.rt_set_option(frame, capi.OPT_PIPELINES, args.pipelines) == 0
    if args.samples_in_flight:
        assert lib.rt_set_option(frame, capi.OPT_SAMPLES_IN_FLIGHT, args.samples_in_flight) == 0
    if args.shade_partition is not None:
        assert lib.rt_set_option(frame, capi.OPT_SHADE_PARTITION, args.shade_partition) == 0
    if args.trace_tune:
        assert lib.rt_set_option(frame, capi.OPT_TRACE_TUNE, args.trace_tune) == 0
    if args.overlap_shadow is not None:
        assert lib.rt_set_option(frame, capi.OPT_OVERLAP_SHADOW, args.overlap_shadow) == 0
    if args.trace_waves:
        assert lib.rt_set_option(frame, capi.OPT_TRACE_WAVES, args.trace_waves) == 0
    if args.trace_variant is not None:
        assert lib.rt_set_option(frame, capi.OPT_TRACE_VARIANT, args.trace_variant) == 0
    if args.path_state_gb > 0:
        assert lib.rt_set_option(frame, capi.OPT_PATH_STATE_LIMIT_MB, int(args.path_state_gb * 1024)) == 0
    if args.small_launch_paths is not None:
        assert lib.rt_set_option(frame, capi.OPT_SMALL_LAUNCH_PATHS, args.small_launch_paths) == 0
    if args.compact_log is not None:
        assert lib.rt_set_option(frame, capi.OPT_COMPACT_LOG, args.compact_log) == 0
    if args.tail_lanes is not None:
        assert lib.rt_set_option(frame, capi.OPT_TRACE_TAIL_LANES, args.tail_lanes) == 0
    if args.chunk_refill is not None:
        assert lib.rt_set_option(frame, capi.OPT_CHUNK_REFILL, args.chunk_refill) == 0
    if args.tail_paths is not None:
        assert lib.rt_set_option(frame, capi.OPT_TRACE_TAIL_PATHS, args.tail_paths) == 0
    if args.per_frame_only:
        if args.frame_kernel is not None:
            assert lib.rt_set_option(frame, capi.OPT_FRAME_KERNEL, args.frame_kernel) == 0
        if args.stage_pipes:
            assert lib.rt_set_option(frame, capi.OPT_STAGE_PIPES, args.stage_pipes) == 0
        pf = per_frame_leg(args, render, lib, frame, capi, max(args.per_frame_frames, 1), ahead=args.samples_ahead)
        if args.moving_camera_frames > 0:
            pf["moving_camera"] = moving_camera_leg(args, render, lib, frame, capi, host, cam, args.moving_camera_frames)
        if rank == 0:
            print(json.dumps(dict(per_frame=pf, config=dict(width=args.width, height=args.height, max_bounces=args.bounces, config=args.config,
                                                            trace_tune=args.trace_tune, small_launch_paths=args.small_launch_paths,
                                                            trace_variant=args.trace_variant, overlap_shadow=args.overlap_shadow))), flush=True)
        return 0
    t_alloc0 = time.perf_counter()
    in_flight = render.reserve_samples(max(spp_timed, spp_warm))
    render.finish()
    t_first_alloc = time.perf_counter() - t_alloc0          # the process's first allocation of the batch's per-path buffers (hipMalloc of up to ~100 GB)

    # ---- warm-up ------------------------------------------------------------
    t_warm0 = time.perf_counter()
    render.render_samples(spp_warm) if spp_warm > 0 else None
    render.finish()
    t_warm = time.perf_counter() - t_warm0                # with bit 1 of --adaptive-fold: probe + worker + adoption are in here
    fold_share = None
    if share_folds:
        # rank 0's records (adapted inside its warm-up: bit 1 waits) to everybody, then one more untimed step on the folds the timed region runs on
        t_s0 = time.perf_counter()
        ctx_h = host.load().rth_render_ctx_handle(render.handle)
        meta = torch.zeros(4, dtype=torch.int64)
        if rank == 0:
            cl, sh, ent = capi.export_folds(ctx_h)
            meta[:] = torch.tensor([len(cl), len(sh), ent[0], ent[1]])
        dist.broadcast(meta, src=0)
        n_cl, n_sh = int(meta[0]), int(meta[1])
        t_cl = torch.from_numpy(cl) if rank == 0 else torch.empty((n_cl, 64), dtype=torch.uint8)
        t_sh = torch.from_numpy(sh) if rank == 0 else torch.empty((max(n_sh, 1), 64), dtype=torch.uint8)
        if rank == 0 and n_sh == 0:
            t_sh = torch.zeros((1, 64), dtype=torch.uint8)
        dist.broadcast(t_cl, src=0)
        dist.broadcast(t_sh, src=0)
        if rank != 0:
            capi.import_folds(ctx_h, t_cl.numpy(), t_sh.numpy()[:n_sh], (int(meta[2]), int(meta[3])))
        if spp_warm > 0:
            render.render_samples(min(spp_warm, sps))
        render.finish()
        fold_share = dict(records=[n_cl, n_sh], seconds=round(time.perf_counter() - t_s0, 3),
                          what="one fold adaptation per process group: rank 0's 4-wide records (closest-hit + shadow, adapted in its warm-up) broadcast over the launcher's "
                               "gloo group and imported by the other ranks, which uploaded without a shadow tree or an adaptation of their own (untimed)")
    trees_after_warmup = render.tree_report() if args.adaptive_fold else tree_report     # the folds the timed region runs on (later legs may adapt again)
    if args.warmup > 0:     # the gather path too (first use sets up the RCCL channels)
        if group is not None:
            try:
                gather(False)
                sync()
                failed = None
            except Exception as e:                          # noqa: BLE001
                failed = "rt_group_gather_radiance on rank %d: %r" % (rank, e)
            okay = torch.tensor([0 if failed else 1], dtype=torch.int32)
            dist.all_reduce(okay, op=dist.ReduceOp.MIN)
            if int(okay.item()) == 0:                       # same agreement as at start-up: everybody falls back to gloo
                reasons = [None] * world
                dist.all_gather_object(reasons, failed)
                rccl_fallback = next((r for r in reasons if r), "the gather failed on another rank")
                group.close()
                group = None
        if group is None:
            gather(False)
    sync()
    # the timed region starts from a reset accumulation (sample indices 0..K-1, counters at 0)
    assert lib.rt_reset(frame) == 0
    st0 = render.stats()
    lib.rt_set_option(frame, capi.OPT_PROFILE, 0)          # nothing but the work inside the timed region: no event pairs around the launches
    if world > 1:
        dist.barrier()
    sync()

    # ---- timed region: exactly K steps + the one gather ----------------------
    t0 = time.perf_counter()
    render.render_samples(spp_timed)
    render.finish()
    t_render = time.perf_counter() - t0
    gather(False)
    sync()
    t_local = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0

    st1 = render.stats()
    closest = st1.closest_rays - st0.closest_rays
    shadow = st1.shadow_rays - st0.shadow_rays
    agg = torch.tensor([float(closest), float(shadow)], dtype=torch.float64)
    tmax = torch.tensor([dt, t_render, t_local - t_render], dtype=torch.float64)
    tmin = torch.tensor([t_render], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
    agg = agg.numpy()
    dt_max = float(tmax[0].item())
    # every rank's own numbers, so that imbalance is attributable (not only min / max)
    mine = dict(rank=rank, render_ms=round(t_render * 1e3, 3), gather_ms=round((t_local - t_render) * 1e3, 3), setup_s=round(t_setup, 2),
                scene_s=round(t_scene, 2), rows=int(local_rows), rays=float(closest + shadow),
                # did the launches stay full as the tile shrank?  samples this rank keeps in flight, and the closest-hit rays of an
                # average launch (bounce 0 carries rows x width x in_flight of them)
                in_flight=int(in_flight), rays_per_launch=round(float(closest) / max(-(-spp_timed // max(int(in_flight), 1)) * (args.bounces + 1), 1), 1),
                rays_in_first_launch=int(local_rows) * args.width * int(in_flight),
                rccl=(list(group.comm_count()) if group is not None else None))
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    full = gather(True)                                      # untimed: the image for the checks below
    # N > 1: the gathered frame against the reference's own kernels (untimed, bounded: 2 samples of the full frame on rank 0's
    # host cores, the same 2 samples by the N tiles, one more gather).  Reported, never fatal: a failure here must not cost
    # the line its measurement.
    tiled_parity = None
    if world > 1 and not args.no_cpu_baseline:
        n_ref, ref_frame, note = [0], None, None
        if rank == 0:
            try:
                from tests import _ref
                if _ref.available():
                    t_ref = time.time()
                    ri = _ref.RefIntegrator(args.width, args.height, render.scene_arrays(), threads=min(16, os.cpu_count() or 1))
                    ri.set_camera(cam)
                    ri.set_max_bounces(args.bounces)
                    ri.integrate(2)
                    ref_frame, n_ref = ri.radiance()[..., :3].copy(), [2]
                    note = "oracle/_ref (the reference's own kernels), 2 spp at %dx%d, %d bounces, %.1f s on rank 0's host" % (
                        args.width, args.height, args.bounces, time.time() - t_ref)
                    del ri
            except Exception as e:                          # noqa: BLE001 -- reported in the line
                note = "reference leg failed: %r" % (e,)
        dist.broadcast_object_list(n_ref, src=0)
        if n_ref[0]:
            try:
                assert lib.rt_reset(frame) == 0
                render.render_samples(n_ref[0])
                sync()
                got_full = gather(True)
                if rank == 0:
                    got = got_full[..., :3].numpy()
                    same = (got == ref_frame) | (np.isnan(got) & np.isnan(ref_frame))
                    tiled_parity = dict(against=note, tiles=world, bit_identical=bool(same.all()), differing_pixels=int((~same.all(-1)).sum()),
                                        what="the frame gathered from the %d tiles (the same collective as the timed one), same samples" % world)
            except Exception as e:                          # noqa: BLE001
                tiled_parity = dict(against=note, tiles=world, error=repr(e))
        elif rank == 0:
            tiled_parity = dict(against=note or "oracle/_ref not built on this box", tiles=world, bit_identical=None)
    # One more step, untimed, with every launch on one stream: in the timed region the shadow trace of bounce b runs
    # beside the closest-hit trace of bounce b + 1 (RT_OPT_OVERLAP_SHADOW), so the launch durations there include the
    # sharing; this step gives the kernels' durations alone on the machine (what the rocprofv3 --pmc passes see).
    def profiled_step(overlap, what):
        """one more step of sps samples, UNTIMED, with HIP-event pairs around every launch (RT_OPT_PROFILE_KERNELS)"""
        assert lib.rt_set_option(frame, capi.OPT_OVERLAP_SHADOW, 1 if overlap else 0) == 0
        lib.rt_set_option(frame, capi.OPT_PROFILE, 1)
        p = capi.rt_profile()
        lib.rt_frame_get_profile(frame, p)                  # drain
        st_a = render.stats()
        render.render_samples(sps)
        render.finish()
        st_b = render.stats()
        lib.rt_frame_get_profile(frame, p)
        lib.rt_set_option(frame, capi.OPT_PROFILE, 0)
        n = max(p.n_trace_closest, 1)
        rays = float(st_b.closest_rays - st_a.closest_rays)
        return dict(what=what, avg_launch_ms=round(p.ms_trace_closest / n, 5), rays_per_launch=round(rays / n, 1),
                    mrays_per_s=round(rays / (p.ms_trace_closest * 1e-3) / 1e6, 1) if p.ms_trace_closest > 0 else 0.0,
                    kernel_ms_per_spp=dict(trace_closest=round(p.ms_trace_closest / sps, 4), trace_shadow=round(p.ms_trace_shadow / sps, 4),
                                           shade=round(p.ms_shade / sps, 4), raygen=round(p.ms_raygen / sps, 4)))

    isolated, live_step = None, None
    overlap_on = args.overlap_shadow is None or args.overlap_shadow != 0
    if world == 1:
        live_step = profiled_step(overlap_on, "one more step of %d spp in the timed region's mode (untimed, instrumented)" % sps)
        if overlap_on:
            # (twice, the faster kept: one of ~20 such steps runs a fifth slower -- clocks after the sustained timed region -- and every
            # ceiling below is priced from this launch duration)
            isolated = min((profiled_step(False, "one more step of %d spp with RT_OPT_OVERLAP_SHADOW = 0 (untimed; the faster of two): each kernel alone "
                                                 "on the machine" % sps) for _ in range(2)), key=lambda p: p["avg_launch_ms"])
            assert lib.rt_set_option(frame, capi.OPT_OVERLAP_SHADOW, 1) == 0

    per_frame = None
    if world == 1 and args.per_frame_frames > 0:
        if args.stage_pipes:
            assert lib.rt_set_option(frame, capi.OPT_STAGE_PIPES, args.stage_pipes) == 0
        per_frame = per_frame_leg(args, render, lib, frame, capi, args.per_frame_frames, ahead=args.samples_ahead)
        per_frame["stage_pipes"] = args.stage_pipes or 1
        if args.frame_kernel is None:
            per_frame["frame_kernel"] = frame_kernel_legs(args, render, lib, frame, capi, per_frame)
        if args.moving_camera_frames > 0:
            per_frame["moving_camera"] = moving_camera_leg(args, render, lib, frame, capi, host, cam, args.moving_camera_frames)
        if args.stage_pipes:
            assert lib.rt_set_option(frame, capi.OPT_STAGE_PIPES, 1) == 0      # the legs below read queues and counters of one pipe

    if rank == 0:
        assert full is not None
        total_rays = agg[0] + agg[1]
        value = total_rays / dt_max / 1e6
        nan_px = int((~torch.isfinite(full[..., :3]).all(-1)).sum().item())
        # CPU legs on a reduced frame of the same scene (bounded, see docstring)
        small_w, small_h = 320, 180          # oracle counters + CPU baseline frame
        arrays = render.scene_arrays()
        per_ray, baseline, ref_img, ref_spp, libm_img = cpu_legs(args, arrays, host.default_camera(small_w, small_h), small_w, small_h, cam)
        parity = tiled_parity
        if ref_img is not None and world == 1:
            # the SAME samples on the GPU (outside the timed region), compared with the reference kernels' image
            assert lib.rt_reset(frame) == 0
            render.render_samples(ref_spp)
            got = render.radiance()[..., :3]
            fin = np.isfinite(ref_img).all(-1) & np.isfinite(got).all(-1)
            num = np.linalg.norm((got[fin].astype(np.float64) - ref_img[fin]).ravel())
            den = np.linalg.norm(ref_img[fin].astype(np.float64).ravel())
            parity = dict(against="oracle/_ref (the reference's own kernels), %d spp at %dx%d, %d bounces" %
                          (ref_spp, args.width, args.height, args.bounces),
                          bit_identical=bool(np.array_equal(got, ref_img, equal_nan=True)), rel_l2=float(num / den) if den > 0 else 0.0,
                          tolerance=1e-4, nan_pixels_ref=int((~np.isfinite(ref_img).all(-1)).sum()),
                          nan_pixels_hip=int((~np.isfinite(got).all(-1)).sum()),
                          differing_pixels=int((~((got == ref_img) | (np.isnan(got) & np.isnan(ref_img))).all(-1)).sum()))
            parity["median_pixel_rel_err"] = median_pixel_rel_err(got, ref_img)
            if libm_img is not None:
                # the reference against ITSELF: its own kernels over the pinned builtins (libref.so) and over glibc libm (libref_libm.so), same frame,
                # same samples, no GPU involved -- what "the reference's result" moves by when only its builtin library changes.  The HIP path equals
                # the first bit for bit, so its distance to the second is this number.
                finr = np.isfinite(libm_img).all(-1) & np.isfinite(ref_img).all(-1)
                numr = np.linalg.norm((ref_img[finr].astype(np.float64) - libm_img[finr]).ravel())
                denr = np.linalg.norm(libm_img[finr].astype(np.float64).ravel())
                parity["reference_self_rel_l2"] = float(numr / denr) if denr > 0 else 0.0
                parity["reference_self_median_pixel_rel_err"] = median_pixel_rel_err(ref_img, libm_img)
                parity["reference_self_p99_pixel_rel_err"] = p99_pixel_rel_err(ref_img, libm_img)
                parity["reference_self"] = ("oracle/_ref/libref.so vs oracle/_ref/libref_libm.so (the reference's unmodified kernels over two conformant builtin "
                                            "libraries), the CPU leg's frame: a property of the reference, measured without the GPU")
                parity["median_pixel_rel_err_vs_libm_build"] = median_pixel_rel_err(got, libm_img)
                finl = np.isfinite(libm_img).all(-1) & np.isfinite(got).all(-1)
                numl = np.linalg.norm((got[finl].astype(np.float64) - libm_img[finl]).ravel())
                denl = np.linalg.norm(libm_img[finl].astype(np.float64).ravel())
                parity["rel_l2_vs_libm_build"] = float(numl / denl) if denl > 0 else 0.0
                parity["libm_build"] = ("oracle/_ref/libref_libm.so: the same reference kernels with glibc libm builtins instead of "
                                        "raytracing_amd/csrc/rt_detmath.h, same frame and samples; %d pixels differ" %
                                        int((~((got == libm_img) | (np.isnan(got) & np.isnan(libm_img))).all(-1)).sum()))
                parity["libm_build_within_tolerance"] = bool(parity["rel_l2_vs_libm_build"] < 1e-4)     # reported, not asserted: the bit-exact pin is libref.so
            assert parity["rel_l2"] < 1e-4 or args.closest_tree, "radiance differs from the reference kernels: %r" % parity
            if (args.libm_series or (args.libm_series is None and args.config == 5)) and libm_img is not None:
                # How the distance to the libm build falls with the sample count (tools/libm_tolerance_series.py): a reduced
                # frame of the same scene, the same sample indices on both sides, bounded to a few seconds of host time
                try:
                    import importlib.util
                    spec = importlib.util.spec_from_file_location("libm_tolerance_series", os.path.join(ROOT, "tools", "libm_tolerance_series.py"))
                    lts = importlib.util.module_from_spec(spec)
                    spec.loader.exec_module(lts)
                    sw, sh = 960, 540
                    r2 = host.Render(sw, sh, scene)
                    cam2 = host.default_camera(sw, sh)
                    r2.set_camera(cam2)
                    r2.set_max_bounces(args.bounces)
                    state = dict(done=0)
                    def hip_image_at(n):
                        r2.render_samples(n - state["done"]); state["done"] = n
                        return r2.radiance()
                    spps = [int(x) for x in (args.libm_series or "1,2,4,8").split(",")]
                    pts = lts.series(arrays, sw, sh, args.bounces, spps, cam2, hip_image_at, min(64, os.cpu_count() or 1))
                    slope, c1, cross = lts.fit([p["spp"] for p in pts], [p["rel_l2"] for p in pts])
                    parity["rel_l2_vs_libm_build_series"] = dict(frame="%dx%d of the same scene, %d bounces, sample indices 0..n-1 on both sides" % (sw, sh, args.bounces),
                                                                 points=pts, fitted_slope=slope, fitted_rel_l2_at_1_spp=c1, crosses_1e_4_at_spp=cross,
                                                                 is_also="the reference's own sensitivity to its builtin library: the HIP path is bit-identical to libref.so "
                                                                         "(parity.bit_identical on the full frame), so each point is libref.so vs libref_libm.so as well",
                                                                 longer_series="profiles/r04_libm_tolerance_series_cfg5.json (2 .. 128 spp, tools/libm_tolerance_series.py)")
                    r2.close()
                except Exception as e:                      # noqa: BLE001 -- reported, never fatal to the measurement
                    parity["rel_l2_vs_libm_build_series"] = dict(error=repr(e))
            if args.config == 1 and world == 1:
                # BASELINE configs[0] has published per-sample counts (SURVEY.md 8d "Config 1" / Appendix A: the reference's own
                # Scene + Bvh + unmodified kernels over glibc libm, sample index 0): the GPU's queue counters for that sample
                assert lib.rt_reset(frame) == 0
                render.render_samples(1)
                s1 = render.stats()
                parity["config_1_sample_0"] = dict(
                    closest_rays=int(s1.closest_rays), shadow_rays=int(s1.shadow_rays),
                    active_per_bounce=[int(x) for x in s1.last_active[:args.bounces + 1]], shadow_per_bounce=[int(x) for x in s1.last_shadow[:args.bounces + 1]],
                    survey_appendix_a=dict(closest_rays=265979, shadow_rays=147128, active_per_bounce=SURVEY_A_ACTIVE, shadow_per_bounce=SURVEY_A_SHADOW),
                    equals_survey=bool(int(s1.closest_rays) == 265979 and int(s1.shadow_rays) == 147128 and
                                       [int(x) for x in s1.last_active[:5]] == SURVEY_A_ACTIVE and [int(x) for x in s1.last_shadow[:5]] == SURVEY_A_SHADOW)
                    if (args.width, args.height, args.bounces) == (256, 256, 4) else None)
        else:
            # NaN pixels are legal in the reference arithmetic (inf * 0 in the mirror branch, material.h:79-81,230: coarse
            # mirror spheres produce them by the hundred, and the reference produces the same ones -- `parity` compares them
            # when the CPU leg runs); reported as config.non_finite_pixels, and a frame FULL of them is a bug
            assert nan_px <= 1e-2 * args.width * args.height, "too many non-finite pixels: %d" % nan_px
        roofline = roofline_object(args, world, live_step, per_ray, isolated)
        if isolated is not None:
            roofline["live_isolated"] = isolated
        trees_now = trees_after_warmup.strip().split("\n")
        # The headline runs on the fold ADAPTED to this view (RT_CTX_OPT_ADAPTIVE_FOLD, bit 1: the warm-up waits for it).  Beside it: what the
        # adaptation took, and the same job on the fold rt_scene_upload makes (surface area) -- the scene uploaded again with the option off, a
        # warm-up step and a few timed ones, after everything else.
        adaptation = None
        if args.adaptive_fold:
            import re
            m = re.search(r"([0-9.]+) s on a worker thread", "\n".join(trees_now))
            steady = dt_max / max(args.steps, 1)
            adaptation = dict(worker_s=float(m.group(1)) if m else None, warmup_s=round(t_warm, 3), steady_step_s=round(steady, 4),
                              seconds_to_adapted=round(max(t_warm - args.warmup * steady, float(m.group(1)) if m else 0.0), 3) if (args.adaptive_fold & 2) else None,
                              note="seconds_to_adapted: the first rt_integrate's probe + the worker thread's folds + the upload, as the warm-up saw them "
                                   "(bit 1: it waits; the library default adopts whenever the worker is done and renders on the upload's fold meanwhile)")
        surface_area_fold = None
        if args.adaptive_fold and world == 1 and args.surface_area_fold_steps > 0 and not args.per_frame_only:
            try:
                render.set_adaptive_fold(0)                      # ... and uploads again
                sa_steps = max(args.surface_area_fold_steps, min(args.steps, int(round(0.4 / max(dt_max / max(args.steps, 1), 1e-6)))))   # >= 0.4 s of work
                # (the buffers of the WHOLE job's batches before the timer, as the headline has them: the per-frame legs have shrunk them meanwhile, and a
                # hipMalloc of ~100 GB takes seconds -- round 6's first lines timed it here for configs 2 and 5, whose step is smaller than their batch)
                render.reserve_samples(max(sa_steps * sps, spp_timed))
                render.render_samples(sps); render.finish()
                assert lib.rt_reset(frame) == 0
                sa0 = render.stats()
                ta = time.perf_counter()
                render.render_samples(sa_steps * sps); render.finish()
                tb = time.perf_counter() - ta
                sa1 = render.stats()
                sa_rays = float((sa1.closest_rays - sa0.closest_rays) + (sa1.shadow_rays - sa0.shadow_rays))
                surface_area_fold = dict(value=round(sa_rays / tb / 1e6, 2), unit="Mrays/s", steps=sa_steps,
                                         what="the same job on the fold rt_scene_upload makes (RT_CTX_OPT_ADAPTIVE_FOLD = 0), same box, untimed by the driver")
            except Exception as e:                              # noqa: BLE001 -- reported, never fatal to the measurement
                surface_area_fold = dict(error=repr(e))
        if cold_job and "rays" in cold_job:
            # (the growth from the cold job's buffers to the measured job's full batch is path_state_alloc_s: what a job that reserves the full batch pays before its first ray)
            cold_job["full_batch"] = dict(samples_in_flight=int(in_flight), alloc_s=round(t_first_alloc, 3),
                                          what="growing the buffers from the cold job's to the measured job's full batch afterwards (hipFree + hipMalloc), untimed by either")
            cold_job["over_the_warm_headline"] = round(cold_job["mrays_per_s_render"] / value, 4) if value > 0 else None
        scaling_estimate = None
        est_path = os.path.join(ROOT, "profiles", "r06_tile_efficiency.json")
        if not os.path.exists(est_path):
            est_path = os.path.join(ROOT, "profiles", "r05_tile_efficiency.json")
        if os.path.exists(est_path) and args.config == 4:
            try:
                scaling_estimate = json.load(open(est_path))
                scaling_estimate["measured"] = False
                # what the tile timing leaves out: the one gather.  Every rank sends its tile (16 B per pixel) over its OWN xGMI link to the root
                # (point-to-point links, ~153 GB/s each: /opt/skills/guides/MI355X_MICROARCH.md), so the transfers run side by side
                per_rank_mb = args.width * args.height * 16 / 8 / 1e6
                scaling_estimate["gather_model_ms"] = dict(n8=round(per_rank_mb / 153e3 * 1e3 + 0.03, 3),
                                                           formula="N = 8: %.2f MB per rank / 153 GB/s per xGMI link, seven links into the root in parallel, + ~0.03 ms launch and "
                                                                   "synchronisation = a model (no multi-GPU node has been available); against 56 ms of rendering per rank at 256 spp" % per_rank_mb)
            except Exception as e:                              # noqa: BLE001
                scaling_estimate = dict(error=repr(e), measured=False)
        name, cus, mem = render_ctx_info(capi, host, render)
        if world == 1:
            gather_info = dict(transport="none (single tile, device copy)", ms=round(float(tmax[2].item()) * 1e3, 3), nranks=1)
        else:
            rccl_counts = sorted({r["rccl"][0] for r in per_rank if r["rccl"]})
            gather_info = dict(transport=("gloo over host memory (--debug-shared-gpu)" if rccl_fallback is None else
                                          "gloo over host memory -- RCCL FALLBACK: %s" % rccl_fallback) if group is None else
                               "RCCL ncclGather over xGMI (rt_group_gather_radiance)", nranks=world,
                               rccl_nranks=(rccl_counts[0] if len(rccl_counts) == 1 else rccl_counts) if rccl_counts else None,   # ncclCommCount on every rank
                               rccl_user_ranks=[r["rccl"][1] for r in per_rank] if rccl_counts else None,
                               ms_max=round(float(tmax[2].item()) * 1e3, 3), bytes_per_rank=int(D.max_tile_rows(args.height, world, args.band_height)) * args.width * 16)
        line = dict(metric="Mrays/s (all bounces+shadow)", value=round(value, 2), unit="Mrays/s", n_gpus=world,
                    steps=args.steps, warmup=args.warmup, ms_per_step=round(dt_max * 1e3 / args.steps, 4),
                    ms_per_spp=round(dt_max * 1e3 / spp_timed, 4),
                    higher_is_better=True, scaling="strong", vs_baseline=None, dtype="f32", data="real" if (args.scene or args.config == 1) else "synthetic",
                    config=dict(workload=(("scene file %s (scale %g, flip_yz %d)" % (os.path.basename(args.scene), args.scale, args.flip_yz))
                                          if args.scene else (cfg["name"] % dict(blob=args.blob_tris, ball=args.ball_tris))) +
                                         ", %dx%d, %d-bounce, %d spp per step, default camera, directional light + "
                                         "CGSkies env map" % (args.width, args.height, args.bounces, sps),
                                triangles=int(n_tris), width=args.width, height=args.height,
                                max_bounces=args.bounces, samples_per_step=sps, spp=spp_timed, samples_in_flight=in_flight,
                                path_state_GB=round(st1.path_state_bytes * world / 2 ** 30, 2), chunk_pixels=int(st1.chunk_pixels), pipelines=int(st1.pipelines),
                                tiling="%d interleaved %d-row bands per GPU, 1 gather" % (world, args.band_height)
                                if world > 1 else "single tile",
                                rays_per_step=round(total_rays / args.steps, 1), non_finite_pixels=nan_px,
                                stack_spill_lane_steps=int(st1.stack_spills), rays_left_to_the_bvh2_kernel=int(st1.slow_rays),
                                log_inline_entries=int(st1.log_inline_entries), log_fallbacks=int(st1.log_fallbacks),   # 0 inline = the full log layout
                                trees=trees_now, adaptive_fold=args.adaptive_fold,    # what rt_scene_upload measured when it chose the shadow (/ closest-hit) tree
                                setup_s=round(t_setup, 2), scene_s=round(t_scene, 2),     # scene_s: parse / generate (or load the cache); setup_s: + BVH, wide collapse, upload
                                setup_breakdown=setup_breakdown, path_state_alloc_s=round(t_first_alloc, 3),
                                device=name),
                    ranks=dict(render_ms_min=round(float(tmin[0].item()) * 1e3, 3), render_ms_max=round(float(tmax[1].item()) * 1e3, 3),
                               render_ms=[r["render_ms"] for r in per_rank], gather_ms=[r["gather_ms"] for r in per_rank],
                               setup_s=[r["setup_s"] for r in per_rank], setup_s_max=max(r["setup_s"] for r in per_rank),
                               scene_s=[r["scene_s"] for r in per_rank], rows=[r["rows"] for r in per_rank],
                               mrays=[round(r["rays"] / 1e6, 1) for r in per_rank], scene=scene_source, fold_share=fold_share,
                               in_flight=[r["in_flight"] for r in per_rank], rays_per_launch=[r["rays_per_launch"] for r in per_rank],
                               rays_in_first_launch=[r["rays_in_first_launch"] for r in per_rank]),
                    gather=gather_info, per_frame=per_frame, roofline=roofline, parity=parity, cpu_baseline=baseline,
                    adaptation=adaptation, surface_area_fold=surface_area_fold, cold_job=cold_job, scaling_estimate=scaling_estimate)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if group is not None:
        group.close()
    return 0


def render_ctx_info(capi, host, render):
    import ctypes as C
    lib = capi.load()
    ctx = host.load().rth_render_ctx_handle(render.handle)
    name = C.create_string_buffer(256)
    cu = C.c_int()
    mem = C.c_size_t()
    lib.rt_ctx_device_info(ctx, name, 256, C.byref(cu), C.byref(mem))
    return name.value.decode(), cu.value, mem.value


if __name__ == "__main__":
    sys.exit(main())
