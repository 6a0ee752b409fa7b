"""-m gpu: parity WHERE THE METRIC IS QUOTED (VERDICT r01 "what's weak" 1): the stand-ins of BASELINE configs 3, 4
and 5 rendered with the production launch shape -- 128 samples of every pixel in flight, more than 2 M paths per
launch, so that the automatic choice is the persistent wide-tree kernel (k_trace_w4, with k_trace2 behind it) and
path ids, the radiance log and the work distribution run at depth -- compared BIT FOR BIT with the reference's own
kernels (oracle/_ref, RefIntegrator) on the same scene, camera and samples.  Reference side of the comparison:
src/kernels/cl/trace_bvh.cl:144-202, hit_surface.cl:30-186.  Triangle counts are reduced so the CPU reference
finishes in seconds; bench.py repeats the comparison on the full 1080p frame of the full scene (`parity`).

Also here: a tree deep enough that the traversal stack provably spills from LDS to HBM (rt_stats.stack_spills)."""
import os
import tempfile
import numpy as np
import pytest
from tests import _oracle, _ref
from raytracing_amd import capi, host, scenes as S, types as T

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIGHT = ((-0.6, -1.5, 3.5), (15.0, 10.0, 5.0))


def _finish(scene):
    scene.add_directional_light(*LIGHT)
    scene.set_env_path(os.path.join(ROOT, "assets", "ibl", "CGSkies_0036_free.hdr"))
    scene.build_bvh()
    scene.finalize()
    return scene.arrays()


def _config3():
    path = S.shader_balls_obj(tempfile.mkdtemp(prefix="rt_test_"), 6_000)
    return _finish(host.Scene(path)), 3


def _config4():
    return _finish(host.Scene(arrays=S.city_block(300_000))), 8


def _config5():
    return _finish(host.Scene(arrays=S.dense_foliage(400_000))), 16


@pytest.mark.skipif(not _ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("make", [_config3, _config4, _config5], ids=["config3_shader_balls", "config4_city_block", "config5_dense_foliage"])
def test_production_launch_shape_matches_the_reference_kernels_bit_for_bit(make):
    arrays, bounces = make()
    w, h, spp = 192, 108, 128
    assert w * h * spp >= 2_000_000                      # a production-sized batch: > 2 M paths per launch at bounce 0
    cam = T.default_camera(w, h)
    ctx = capi.Context(0)
    ctx.upload_scene(arrays)
    fr = capi.Frame(ctx, w, h)
    fr.set_camera(cam)
    fr.set_max_bounces(bounces)
    assert fr.reserve_samples(spp) == spp
    fr.integrate(spp)                                    # ONE batch: 128 samples of every pixel travel together
    got = fr.radiance()[..., :3]
    st = fr.stats()
    assert st.samples_in_flight == spp and st.last_active[0] == w * h * spp
    ri = _ref.RefIntegrator(w, h, arrays, threads=min(32, os.cpu_count() or 1))
    ri.set_camera(cam)
    ri.set_max_bounces(bounces)
    ri.integrate(spp)
    want = ri.radiance()[..., :3]
    diff = ~((got == want) | (np.isnan(got) & np.isnan(want))).all(-1)
    assert not diff.any(), "%d of %d pixels differ, first at %s" % (diff.sum(), diff.size, np.argwhere(diff)[:3].tolist())
    assert (st.closest_rays, st.shadow_rays) == ri.ray_totals()
    # Round 3: launches below RT_OPT_SMALL_LAUNCH_PATHS (3 M rays) run k_trace_w4 in chunk mode -- at this frame size that is
    # every launch of the batch above.  The same batch with that switched off (every launch REFILLING, the regime of the
    # headline's 100 M-ray launches) and with it forced on must give the same bits and counters.
    for small in (0, 4000000000):
        fr.set_option(capi.OPT_SMALL_LAUNCH_PATHS, small)
        fr.reset()
        fr.integrate(spp)
        assert np.array_equal(fr.radiance()[..., :3], got, equal_nan=True), small
        st2 = fr.stats()
        assert (st2.closest_rays, st2.shadow_rays) == ri.ray_totals()
    fr.set_option(capi.OPT_SMALL_LAUNCH_PATHS, 3000000)
    # the same batch through the BVH2 kernels gives the same bits (and the same counters)
    fr.set_option(capi.OPT_TRACE_VARIANT, 8)
    fr.reset()
    fr.integrate(spp)
    assert np.array_equal(fr.radiance()[..., :3], got, equal_nan=True)
    fr.close()
    ctx.close()


@pytest.mark.parametrize("variant,small", [(8, 0), (10, 0), (10, 3000000)], ids=["k_trace2", "k_trace_w4_refilling", "k_trace_w4_chunk_mode"])
def test_deep_tree_spills_the_traversal_stack_to_hbm_and_stays_exact(variant, small, env_map):
    """65 536 large triangles stacked in depth along the view direction: every primary ray overlaps every box, so the
    descent to the first leaf pushes one far child per BVH2 level (16+ levels) -- more than the 10 / 12 entries the
    kernels keep in LDS.  rt_stats.stack_spills proves the HBM spill path ran; the image equals the oracle's."""
    n = 65536
    k = np.arange(n, dtype=np.float32)
    y = 1.0 + k * np.float32(1.0 / 4096.0)
    P = np.zeros((n, 3, 3), np.float32)
    P[:, 0] = np.stack([np.full(n, -9.0, np.float32), y, np.full(n, -8.0, np.float32)], 1)
    P[:, 1] = np.stack([np.full(n, 9.0, np.float32), y, np.full(n, -8.0, np.float32)], 1)
    P[:, 2] = np.stack([np.zeros(n, np.float32), y, np.full(n, 12.0, np.float32)], 1)
    N = np.tile(np.array([0, -1, 0], np.float32), (n, 3, 1))
    tris = S.to_triangles([(P, N, np.zeros((n, 3, 2), np.float32), 0)])
    mats = np.array([S.make_material(kd=(0.6, 0.6, 0.6), ks=(0.2, 0.2, 0.2), roughness=0.5)], dtype=T.packed_material)
    s = host.Scene(arrays=dict(triangles=tris, materials=mats))
    s.add_directional_light(*LIGHT)
    s.build_bvh(); s.set_env_image(env_map); s.finalize()
    arrays = s.arrays()
    w, h, b, spp = 48, 32, 2, 2
    cam = T.default_camera(w, h)
    ctx = capi.Context(0)
    ctx.upload_scene(arrays)
    fr = capi.Frame(ctx, w, h)
    fr.set_camera(cam); fr.set_max_bounces(b)
    fr.set_option(capi.OPT_TRACE_VARIANT, variant)
    fr.set_option(capi.OPT_SMALL_LAUNCH_PATHS, small)
    fr.integrate(spp)
    st = fr.stats()
    assert st.stack_spills > 0
    orc = _oracle.Oracle(w, h, arrays)
    orc.set_camera(cam); orc.set_max_bounces(b); orc.integrate(spp)
    assert np.array_equal(fr.radiance()[..., :3], orc.radiance()[..., :3], equal_nan=True)
    assert (st.closest_rays, st.shadow_rays) == orc.ray_totals()
    fr.close()
    ctx.close()


def test_shrinking_the_batch_on_device_memory_pressure_keeps_the_sum_exact(golden_scenes):
    """rt_integrate itself triggers the 'halve the batch until it fits' fallback (no reservation first);
    RT_OPT_DEBUG_ALLOC_LIMIT makes allocations above 3 samples in flight fail like a full device."""
    w, h, b, spp = 64, 48, 5, 11
    sc = golden_scenes["coverage"]
    cam = T.default_camera(w, h)
    ctx = capi.Context(0)
    ctx.upload_scene(sc)
    base = capi.Frame(ctx, w, h)
    base.set_camera(cam); base.set_max_bounces(b); base.set_option(capi.OPT_SAMPLES_IN_FLIGHT, 1)
    base.integrate(spp)
    fr = capi.Frame(ctx, w, h)
    fr.set_camera(cam); fr.set_max_bounces(b)
    fr.set_option(capi.OPT_DEBUG_ALLOC_LIMIT, 3)
    fr.integrate(spp)                                   # asks for 11 in flight, gets 2 (11 -> 5 -> 2), must not fail
    st = fr.stats()
    assert st.samples_in_flight <= 3 and st.samples_in_flight_limit == st.samples_in_flight
    assert fr.sample_count() == spp
    assert np.array_equal(fr.radiance(), base.radiance(), equal_nan=True)
    fr.close(); base.close(); ctx.close()


@pytest.mark.parametrize("variant", [5, 10])
def test_bounded_path_state_renders_the_tile_in_chunks_bit_identically(variant, golden_scenes):
    """RT_OPT_PATH_STATE_LIMIT_MB: the per-path buffers (ray queues + radiance log) are capped and every batch of
    samples runs the wavefront loop chunk by chunk over the tile's pixels.  Same bits, same counters."""
    w, h, b, spp = 160, 100, 6, 9
    sc = golden_scenes["coverage"]
    cam = T.default_camera(w, h)
    ctx = capi.Context(0)
    ctx.upload_scene(sc)
    base = capi.Frame(ctx, w, h)
    base.set_camera(cam); base.set_max_bounces(b); base.set_option(capi.OPT_SAMPLES_IN_FLIGHT, 4)
    base.integrate(spp)
    bs = base.stats()
    assert bs.chunk_pixels == w * h
    for tile in (dict(), dict(tile_rank=1, tile_count=2, band_height=4)):
        ref = capi.Frame(ctx, w, h, **tile)
        ref.set_camera(cam); ref.set_max_bounces(b); ref.set_option(capi.OPT_SAMPLES_IN_FLIGHT, 4); ref.integrate(spp)
        fr = capi.Frame(ctx, w, h, **tile)
        fr.set_camera(cam); fr.set_max_bounces(b)
        fr.set_option(capi.OPT_SAMPLES_IN_FLIGHT, 4)
        fr.set_option(capi.OPT_TRACE_VARIANT, variant)
        fr.set_option(capi.OPT_PATH_STATE_LIMIT_MB, 8)       # 4 x 4096 pixels x ~460 B: 4096-pixel chunks
        fr.integrate(spp)                                    # 4 + 4 + 1 samples, every batch in 2..4 chunks
        st, rs = fr.stats(), ref.stats()
        assert st.chunk_pixels == 4096 and st.path_state_bytes <= 8 << 20
        assert np.array_equal(fr.radiance(), ref.radiance(), equal_nan=True)
        assert (st.closest_rays, st.shadow_rays) == (rs.closest_rays, rs.shadow_rays)
        assert list(st.last_active[: b + 1]) == list(rs.last_active[: b + 1])      # chunks of the last batch add up
        # lifting the limit afterwards keeps accumulating exactly
        fr.set_option(capi.OPT_PATH_STATE_LIMIT_MB, 0)
        fr.integrate(3); ref.integrate(3)
        assert np.array_equal(fr.radiance(), ref.radiance(), equal_nan=True)
        fr.close(); ref.close()
    base.close(); ctx.close()


def test_compact_radiance_log_and_its_fallback_are_bit_identical():
    """RT_OPT_COMPACT_LOG = 1 (round 3, opt-in): batches of >= 8 samples in flight keep six inline log entries per path plus bump-allocated
    overflow blocks for an eighth of the paths instead of the 2 (B + 1) worst case.  Same bits and counters as the full layout
    and as the oracle, with a third less path state; a pool that runs dry (here: shrunk to 64 blocks by the test hook) makes
    rt_integrate discard the batch and repeat it in the full layout -- once, the frame then stays there -- still the same bits."""
    w, h, b, spp = 96, 64, 8, 24
    sc = _finish(host.Scene(arrays=S.city_block(40_000)))     # an open scene: a few per cent of its paths outgrow six entries
    cam = T.default_camera(w, h)
    orc = _oracle.Oracle(w, h, sc)
    orc.set_camera(cam); orc.set_max_bounces(b); orc.integrate(spp)
    want = orc.radiance()[..., :3]
    ctx = capi.Context(0)
    ctx.upload_scene(sc)
    results = {}
    for name, opts in (("full", {capi.OPT_COMPACT_LOG: 0}), ("compact", {capi.OPT_COMPACT_LOG: 1}),
                       ("fallback", {capi.OPT_COMPACT_LOG: 1, capi.OPT_DEBUG_LOG_POOL_DIV: 1000000000}),
                       ("compact_chunked", {capi.OPT_PATH_STATE_LIMIT_MB: 12})):     # the default: compact when the caller bounds the state
        fr = capi.Frame(ctx, w, h)
        fr.set_camera(cam); fr.set_max_bounces(b)
        fr.set_option(capi.OPT_SAMPLES_IN_FLIGHT, 8)
        for k, v in opts.items():
            fr.set_option(k, v)
        fr.integrate(spp)                                    # three batches of 8
        st = fr.stats()
        got = fr.radiance()[..., :3]
        assert np.array_equal(got, want, equal_nan=True), name
        assert (st.closest_rays, st.shadow_rays) == orc.ray_totals(), name
        results[name] = (st.log_inline_entries, st.log_fallbacks, st.path_state_bytes, st.chunk_pixels)
        fr.close()
    assert results["full"][:2] == (0, 0) and results["compact"][:2] == (6, 0), results
    assert results["compact"][2] < 0.75 * results["full"][2]                   # 290 vs 412 bytes per path at 8 bounces
    assert results["fallback"][:2] == (0, 1)                                  # one batch repeated, then the full layout for good
    assert results["fallback"][2] == results["full"][2] and results["fallback"][3] == w * h     # ... as if it had started there
    assert results["compact_chunked"][0] == 6 and results["compact_chunked"][3] < w * h
    ctx.close()


def test_stage_api_after_a_chunked_batch_renders_the_whole_tile(golden_scenes):
    """with RT_OPT_PATH_STATE_LIMIT_MB set, an rt_integrate with several samples in flight leaves the
    per-path buffers cut into chunks of pixels; a stage-API sample afterwards (the reference's Integrate() through the hooks:
    rt_generate_rays ... rt_advance_sample) must render the WHOLE tile -- one sample of it fits -- not the first chunk only
    (round 2 advanced the sample count over a partial image without an error).  Same for the per-frame features."""
    w, h, b = 160, 100, 4
    sc = golden_scenes["coverage"]
    cam = T.default_camera(w, h)
    ctx = capi.Context(0)
    ctx.upload_scene(sc)
    fr = capi.Frame(ctx, w, h)
    fr.set_camera(cam); fr.set_max_bounces(b)
    fr.set_option(capi.OPT_SAMPLES_IN_FLIGHT, 4)
    fr.set_option(capi.OPT_PATH_STATE_LIMIT_MB, 8)
    fr.integrate(4)
    assert fr.stats().chunk_pixels < w * h                       # the batch really ran in chunks
    def stage_sample():
        fr.generate_rays()
        for bounce in range(b + 1):
            fr.intersect(bounce); fr.shade(bounce); fr.intersect_shadow(bounce)
        fr.advance_sample()
    stage_sample()
    assert fr.stats().chunk_pixels == w * h                      # re-allocated for one sample of the whole tile
    stage_sample()
    orc = _oracle.Oracle(w, h, sc)
    orc.set_camera(cam); orc.set_max_bounces(b); orc.integrate(6)
    assert fr.sample_count() == 6
    assert np.array_equal(fr.radiance()[..., :3], orc.radiance()[..., :3], equal_nan=True)
    st = fr.stats()
    assert (st.closest_rays, st.shadow_rays) == orc.ray_totals()
    # ... and rt_integrate with a per-frame feature (the AOV viewer needs the whole tile in one chunk) after chunked batches
    fr.integrate(4)
    assert fr.stats().chunk_pixels < w * h
    fr.set_option(capi.OPT_AOV, 3)
    fr.integrate(1)
    assert fr.stats().chunk_pixels == w * h and fr.sample_count() == 11
    fr.close(); ctx.close()


def test_stage_pipes_carry_one_sample_as_chunks_on_streams_of_their_own(golden_scenes):
    """RT_OPT_STAGE_PIPES (round 5): ONE sample per pixel in flight -- the reference's frame-by-frame pattern through the stage API,
    and rt_integrate(f, 1) -- travels as 2 .. 4 chunks of the tile, each on a pipe (stream + per-path buffers) of its own, so that the
    chunks' launch tails overlap.  Same radiance and ray counters bit for bit as the one-chunk frame; larger batches afterwards (another
    allocation), the stage API again, the presented image, and the debug readers' refusal."""
    w, h, b = 640, 416, 3                                    # >= 512 x 512 pixels: smaller tiles stay in one chunk
    sc = golden_scenes["coverage"]
    cam = T.default_camera(w, h)
    ctx = capi.Context(0)
    ctx.upload_scene(sc)
    one = capi.Frame(ctx, w, h)
    one.set_camera(cam); one.set_max_bounces(b)
    def stage_sample(fr):
        fr.generate_rays()
        for bounce in range(b + 1):
            fr.intersect(bounce); fr.shade(bounce); fr.intersect_shadow(bounce)
        fr.advance_sample()
    for _ in range(3):
        stage_sample(one)
    want3, st3 = one.radiance()[..., :3].copy(), one.stats()
    assert st3.pipelines == 1 and st3.chunk_pixels == w * h
    orc = _oracle.Oracle(w, h, sc)
    orc.set_camera(cam); orc.set_max_bounces(b); orc.integrate(3)
    assert np.array_equal(want3, orc.radiance()[..., :3], equal_nan=True)
    one.integrate(1); one.integrate(4)
    stage_sample(one)
    want9, st9 = one.radiance()[..., :3].copy(), one.stats()
    for pipes in (2, 3, 4):
        fr = capi.Frame(ctx, w, h)
        fr.set_camera(cam); fr.set_max_bounces(b)
        fr.set_option(capi.OPT_STAGE_PIPES, pipes)
        for _ in range(3):
            stage_sample(fr)
        st = fr.stats()
        assert st.pipelines == pipes and st.chunk_pixels < w * h and st.chunk_pixels * pipes >= w * h, (pipes, st.pipelines, st.chunk_pixels)
        assert np.array_equal(fr.radiance()[..., :3], want3, equal_nan=True), pipes
        assert (st.closest_rays, st.shadow_rays) == (st3.closest_rays, st3.shadow_rays), pipes
        assert list(st.last_active[:b + 1]) == list(st3.last_active[:b + 1]) and list(st.last_shadow[:b + 1]) == list(st3.last_shadow[:b + 1]), pipes
        fr.integrate(1)                                      # one sample through rt_integrate: the same chunks
        assert fr.stats().pipelines == pipes
        fr.integrate(4)                                      # a batch: one pipe, the whole tile, four samples in flight
        assert fr.stats().chunk_pixels == w * h
        fr.generate_rays()                                   # ... and back
        with pytest.raises(capi.RtError, match="several pipes"):
            fr.read_queue(0, 0)
        for bounce in range(b + 1):
            fr.intersect(bounce); fr.shade(bounce); fr.intersect_shadow(bounce)
        fr.advance_sample()
        st = fr.stats()
        assert fr.sample_count() == 9 and st.pipelines == pipes
        assert np.array_equal(fr.radiance()[..., :3], want9, equal_nan=True), pipes
        assert (st.closest_rays, st.shadow_rays) == (st9.closest_rays, st9.shadow_rays), pipes
        assert np.array_equal(fr.resolve()[..., :3], one.resolve()[..., :3], equal_nan=True), pipes
        fr.close()
    one.close(); ctx.close()


def test_pipelined_chunks_on_several_streams_are_bit_identical(golden_scenes):
    """RT_OPT_PIPELINES: a large batch (>= 4 M paths) is cut into chunks that travel through the wavefront loop on
    separate pipes (per-path buffers + HIP stream each) so that launch tails overlap.  1, 2, 3 and 4 pipes, with and
    without a memory limit that makes more chunks than pipes: same bits, same counters."""
    w, h, b, spp = 256, 160, 3, 128 + 40                 # 5.2 M paths per batch; a second, partial batch
    sc = golden_scenes["coverage"]
    cam = T.default_camera(w, h)
    ctx = capi.Context(0)
    ctx.upload_scene(sc)
    results = []
    for pipes, limit_mb in ((1, 0), (2, 0), (3, 0), (4, 0), (2, 512), (3, 700)):
        fr = capi.Frame(ctx, w, h)
        fr.set_camera(cam); fr.set_max_bounces(b)
        fr.set_option(capi.OPT_PIPELINES, pipes)
        fr.set_option(capi.OPT_SAMPLES_IN_FLIGHT, 128)
        if limit_mb:
            fr.set_option(capi.OPT_PATH_STATE_LIMIT_MB, limit_mb)
        fr.integrate(spp)
        st = fr.stats()
        assert st.pipelines == pipes
        assert capi.Frame(ctx, 8, 8).stats().pipelines == 1          # default: one pipe
        if limit_mb:
            assert st.path_state_bytes <= limit_mb << 20 and st.chunk_pixels * pipes < w * h      # more chunks than pipes
        results.append((fr.radiance().copy(), st.closest_rays, st.shadow_rays, list(st.last_active[: b + 1])))
        fr.close()
    for r in results[1:]:
        assert np.array_equal(r[0], results[0][0], equal_nan=True)
        assert r[1:] == results[0][1:]
    orc = _oracle.Oracle(w, h, sc)
    orc.set_camera(cam); orc.set_max_bounces(b); orc.integrate(2)
    fr = capi.Frame(ctx, w, h)
    fr.set_camera(cam); fr.set_max_bounces(b); fr.set_option(capi.OPT_SAMPLES_IN_FLIGHT, 128); fr.integrate(2)
    assert np.array_equal(fr.radiance()[..., :3], orc.radiance()[..., :3], equal_nan=True)
    fr.close(); ctx.close()


def test_shadow_trace_on_the_side_stream_and_launch_timeline(golden_scenes):
    """RT_OPT_OVERLAP_SHADOW: inside rt_integrate the shadow trace of bounce b runs on a second stream beside the
    closest-hit trace and k_shade of bounce b + 1 (double-buffered shadow queue, work heads per flavour).  On and off,
    alone and combined with several pipes / a memory limit / the stage API in between: same bits, same counters.
    Also rt_frame_debug_timeline (tools/launch_timeline.py): the instrumented kernel instance changes nothing and its
    record is plausible."""
    import ctypes as C
    w, h, b, spp = 256, 160, 4, 128 + 24                 # 5.2 M paths per batch (k_trace_w4), a second, partial batch
    sc = golden_scenes["coverage"]
    cam = T.default_camera(w, h)
    ctx = capi.Context(0)
    ctx.upload_scene(sc)
    lib = capi.load()
    results = []
    for overlap, pipes, limit_mb, timeline in ((1, 1, 0, 0), (0, 1, 0, 0), (1, 2, 0, 0), (1, 1, 600, 0), (0, 2, 600, 0), (1, 1, 0, 1), (0, 1, 0, 1)):
        fr = capi.Frame(ctx, w, h)
        fr.set_camera(cam); fr.set_max_bounces(b)
        fr.set_option(capi.OPT_OVERLAP_SHADOW, overlap)
        fr.set_option(capi.OPT_PIPELINES, pipes)
        fr.set_option(capi.OPT_SAMPLES_IN_FLIGHT, 128)
        if limit_mb:
            fr.set_option(capi.OPT_PATH_STATE_LIMIT_MB, limit_mb)
        if timeline:
            assert lib.rt_frame_debug_timeline(fr.handle, 1, None) == 0
        fr.integrate(spp)
        if timeline:
            out = (C.c_ulonglong * 448)()
            assert lib.rt_frame_debug_timeline(fr.handle, 0, out) == 0
            for bounce in range(b + 1):
                t0, dry, t1, most, ticks, steps = [out[6 * bounce + k] for k in range(6)]
                assert 0 < t0 <= dry <= t1 and t1 - t0 < 100_000_000          # a launch takes well under a second
                assert 0 < steps <= most < 4096 and 0 < ticks <= t1 - t0
            assert out[6 * (b + 1) + 2] == 0                                  # nothing ran beyond max_bounces
            assert sum(out[384 + i] for i in range(64)) > 0                   # waves were counted leaving
        st = fr.stats()
        results.append((fr.radiance().copy(), st.closest_rays, st.shadow_rays, list(st.last_active[: b + 1])))
        fr.close()
    for r in results[1:]:
        assert np.array_equal(r[0], results[0][0], equal_nan=True)
        assert r[1:] == results[0][1:]
    # the stage API (everything on one stream) between two rt_integrate calls of a frame that overlaps
    fr = capi.Frame(ctx, w, h)
    fr.set_camera(cam); fr.set_max_bounces(b); fr.set_option(capi.OPT_SAMPLES_IN_FLIGHT, 128)
    ref = capi.Frame(ctx, w, h)
    ref.set_camera(cam); ref.set_max_bounces(b); ref.set_option(capi.OPT_SAMPLES_IN_FLIGHT, 128); ref.set_option(capi.OPT_OVERLAP_SHADOW, 0)
    for f in (fr, ref):
        f.integrate(130)
        f.generate_rays()
        for bounce in range(b + 1):
            f.intersect(bounce); f.shade(bounce); f.intersect_shadow(bounce)
        f.advance_sample()
        f.integrate(3)
    assert np.array_equal(fr.radiance(), ref.radiance(), equal_nan=True)
    orc = _oracle.Oracle(w, h, sc)
    orc.set_camera(cam); orc.set_max_bounces(b); orc.integrate(2)
    f2 = capi.Frame(ctx, w, h)
    f2.set_camera(cam); f2.set_max_bounces(b); f2.set_option(capi.OPT_SAMPLES_IN_FLIGHT, 128); f2.integrate(2)
    assert np.array_equal(f2.radiance()[..., :3], orc.radiance()[..., :3], equal_nan=True)
    for f in (fr, ref, f2):
        f.close()
    ctx.close()
