"""-m gpu: the HIP path (through the C-ABI) against the CPU oracle and the
golden fixtures of the reference build.  Integer/index results and -- because
of the shared arithmetic contract -- fp32 radiance are compared BIT FOR BIT;
the north-star tolerance (rel-L2 < 1e-4) is asserted as well so that the test
states the contract it would fall back to."""
import os
import numpy as np
import pytest
from tests.conftest import GOLDEN_CASES
from tests import _oracle, _ref
from raytracing_amd import capi, host, scenes as S, types as T

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-4   # BASELINE.json north_star: radiance rel-L2 < 1e-4 in fp32


def rel_l2(a, b):
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    name, cus, mem = c.device_info()
    assert "gfx950" in name and cus == 256, name
    yield c
    c.close()


def render(ctx, scene, w, h, cam, bounces, spp, furnace=False, slots=None, **tile):
    ctx.upload_scene(scene)
    fr = capi.Frame(ctx, w, h, **tile)
    fr.set_camera(cam)
    fr.set_max_bounces(bounces)
    fr.set_option(capi.OPT_WHITE_FURNACE, int(furnace))
    if slots is not None:
        fr.set_option(capi.OPT_SAMPLES_IN_FLIGHT, slots)     # default 0 = auto (several samples per batch)
    fr.integrate(spp)
    return fr


@pytest.mark.parametrize("case", GOLDEN_CASES, ids=[c[0] for c in GOLDEN_CASES])
def test_radiance_matches_reference_golden_vectors(ctx, case, golden_scenes, golden_radiance):
    name, key, w, h, b, spp, furnace = case
    g = golden_radiance
    fr = render(ctx, golden_scenes[key], w, h, g[name + "/camera"], b, spp, furnace, slots=1)
    got = fr.radiance()[..., :3]
    auto = render(ctx, golden_scenes[key], w, h, g[name + "/camera"], b, spp, furnace)   # default batching
    assert np.array_equal(auto.radiance()[..., :3], g[name + "/radiance"])
    assert rel_l2(got, g[name + "/radiance"]) < TOL
    assert np.array_equal(got, g[name + "/radiance"])
    assert np.array_equal(fr.resolve()[..., :3], g[name + "/resolved"])
    st = fr.stats()
    assert (st.closest_rays, st.shadow_rays) == tuple(int(x) for x in g[name + "/totals"])
    assert list(st.last_active[: b + 1]) == list(g[name + "/last_active"])
    assert list(st.last_shadow[: b + 1]) == list(g[name + "/last_shadow"])
    assert fr.sample_count() == spp


def test_device_math_known_answers(ctx):
    """Leaf arithmetic on the device vs the oracle's C functions, bit for bit."""
    lib = _oracle.load()
    rng = np.random.RandomState(11)
    x = (rng.rand(4000) * 6.2831855).astype(np.float32)
    u = rng.rand(4000).astype(np.float32)
    a1 = (rng.rand(4000) * 2 - 1).astype(np.float32)
    a2 = (rng.rand(4000) * 2 - 1).astype(np.float32)
    def cpu(fn, *cols):
        return np.array([fn(*[float(v) for v in row]) for row in zip(*cols)], np.float32)
    bits = lambda v: np.asarray(v, np.float32).view(np.uint32)
    assert np.array_equal(bits(ctx.debug_eval(0, x)), bits(cpu(lib.orc_sinf, x)))
    assert np.array_equal(bits(ctx.debug_eval(1, x)), bits(cpu(lib.orc_cosf, x)))
    assert np.array_equal(bits(ctx.debug_eval(2, u)), bits(cpu(lib.orc_tanf, u)))
    assert np.array_equal(bits(ctx.debug_eval(3, u, np.full_like(u, 2.2))), bits(cpu(lambda v: lib.orc_powf(v, 2.2), u)))
    assert np.array_equal(bits(ctx.debug_eval(3, a1 * 1.5, np.full_like(u, 5.0))),
                          bits(cpu(lambda v: lib.orc_powf(v, 5.0), a1 * 1.5)))
    assert np.array_equal(bits(ctx.debug_eval(4, a1, a2)), bits(cpu(lib.orc_atan2f, a1, a2)))
    assert np.array_equal(bits(ctx.debug_eval(5, a1)), bits(cpu(lib.orc_acosf, a1)))
    assert np.array_equal(bits(ctx.debug_eval(6, u)), bits(np.sqrt(u)))            # IEEE sqrt
    assert np.array_equal(bits(ctx.debug_eval(7, a1, a2)), bits(a1 / a2))          # IEEE divide
    y = (rng.rand(4000) * 0.999).astype(np.float32)
    want = (1.0 / np.sqrt(1.0 + u.astype(np.float64) / (1.0 - y.astype(np.float64)))).astype(np.float32)
    assert np.array_equal(bits(ctx.debug_eval(9, u, y)), bits(want))               # fp64 path of GGX_Sample
    # SampleRandom: (px | py<<16), (sample | dim<<16)
    px, py, smp, dim = rng.randint(0, 1920, 500), rng.randint(0, 1080, 500), rng.randint(0, 256, 500), rng.randint(0, 45, 500)
    a = (px | (py << 16)).astype(np.uint32).view(np.float32)
    b = (smp | (dim << 16)).astype(np.uint32).view(np.float32)
    want = np.array([lib.orc_sample_random(int(p), int(q), int(s), int(d) // 5, int(d) % 5)
                     for p, q, s, d in zip(px, py, smp, dim)], np.float32)
    assert np.array_equal(bits(ctx.debug_eval(8, a, b)), bits(want))


def test_stage_level_parity_rekeyed_by_pixel(ctx, golden_scenes):
    """Per-kernel buffers vs the oracle's.  Queue ORDER is free on the GPU
    (wave-ballot compaction), so queues are compared after sorting by pixel."""
    w, h, bounces = 56, 40, 3
    sc = golden_scenes["coverage"]
    cam = T.default_camera(w, h)
    ctx.upload_scene(sc)
    fr = capi.Frame(ctx, w, h)
    fr.set_camera(cam)
    fr.set_max_bounces(bounces)
    orc = _oracle.Oracle(w, h, sc)
    orc.set_camera(cam)
    orc.stage("reset")
    n = w * h
    fr.generate_rays()
    orc.stage("generate_rays")
    for bounce in range(bounces + 1):
        k = int(orc.buffer("ray_counter%d" % (bounce & 1), np.uint32, 1)[0])
        rays, pix, thr = fr.read_queue(0, bounce)
        assert len(rays) == k
        o_rays = orc.buffer("rays%d" % (bounce & 1), T.ray, n)[:k]
        o_pix = orc.buffer("pixel_indices%d" % (bounce & 1), np.uint32, n)[:k]
        gi, oi = np.argsort(pix), np.argsort(o_pix)
        assert np.array_equal(pix[gi], o_pix[oi])
        assert T.records_equal(rays[gi], o_rays[oi])
        o_thr = orc.buffer("throughputs", T.float3, n)[o_pix[oi]]
        assert T.records_equal(thr[gi], o_thr)
        fr.intersect(bounce)
        orc.stage("intersect", bounce)
        hits = fr.read_hits(k)[gi]
        o_hits = orc.buffer("hits", T.hit, n)[:k][oi]
        assert np.array_equal(hits["primitive_id"], o_hits["primitive_id"])
        m = hits["primitive_id"] != 0xFFFFFFFF
        assert np.array_equal(hits[m].tobytes(), o_hits[m].tobytes())
        fr.shade(bounce)
        for st in ("shade_miss", "clear_counters", "shade_hits"):
            orc.stage(st, bounce)
        ks = int(orc.buffer("shadow_ray_counter", np.uint32, 1)[0])
        srays, spix, sls = fr.read_queue(1, bounce)
        assert len(srays) == ks
        gs, os_ = np.argsort(spix), np.argsort(orc.buffer("shadow_pixel_indices", np.uint32, n)[:ks])
        assert T.records_equal(srays[gs], orc.buffer("shadow_rays", T.ray, n)[:ks][os_])
        o_ls = orc.buffer("direct_light_samples", T.float4, n)[:ks][os_]
        assert T.records_equal(sls[gs], o_ls)
        fr.intersect_shadow(bounce)
        orc.stage("intersect_shadow")
        orc.stage("accumulate")
        assert np.array_equal(fr.radiance()[..., :3].reshape(n, 3),
                              orc.buffer("radiance", np.float32, n * 4).reshape(n, 4)[:, :3])


def test_mid_sample_reads_on_a_compact_log_allocation(ctx):
    """the stage API on an allocation rt_integrate left in the COMPACT log layout (six inline entries + overflow blocks),
    with the radiance read between the stages -- the sum stays the reference's after every bounce, nothing is read before the log.
    Since round 6 the stages themselves run on the FULL layout (rt_generate_rays re-allocates a compact frame once): a compact log
    whose pool runs dry is made good by rt_integrate repeating its batch, and the stage API has nothing it could repeat -- fuzz seed
    5652 of a 10 000-seed campaign (a test-sized pool) came out 20 pixels short.  The second half drives that case: a pool of 64
    blocks, which the batch falls back from and the stages never see."""
    w, h, bounces = 56, 40, 7
    # an OPEN scene (a closed one -- the golden scenes -- runs the overflow pool dry in its first batch and the frame falls back to the
    # full layout, which is not what this test is about)
    scene = host.Scene(arrays=S.city_block(40_000))
    scene.add_directional_light((-0.6, -1.5, 3.5), (15.0, 10.0, 5.0))
    scene.set_env_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets", "ibl", "CGSkies_0036_free.hdr"))
    scene.build_bvh()
    scene.finalize()
    sc = scene.arrays()
    cam = T.default_camera(w, h)
    ctx.upload_scene(sc)
    fr = capi.Frame(ctx, w, h)
    fr.set_camera(cam)
    fr.set_max_bounces(bounces)
    fr.set_option(capi.OPT_SAMPLES_IN_FLIGHT, 8)
    fr.set_option(capi.OPT_COMPACT_LOG, 1)
    fr.integrate(8)
    assert fr.stats().log_inline_entries == 6 and fr.stats().log_fallbacks == 0      # the compact layout is what is allocated
    orc = _oracle.Oracle(w, h, sc)
    orc.set_camera(cam)
    orc.set_max_bounces(bounces)
    orc.integrate(8)
    n = w * h
    for sample in range(2):                                          # two more samples through the fifteen hooks' C calls
        fr.generate_rays()
        orc.stage("generate_rays")
        for bounce in range(bounces + 1):
            fr.intersect(bounce)
            orc.stage("intersect", bounce)
            fr.shade(bounce)
            for st in ("shade_miss", "clear_counters", "shade_hits"):
                orc.stage(st, bounce)
            fr.intersect_shadow(bounce)
            orc.stage("intersect_shadow")
            orc.stage("accumulate")
            assert np.array_equal(fr.radiance()[..., :3].reshape(n, 3), orc.buffer("radiance", np.float32, n * 4).reshape(n, 4)[:, :3]), (sample, bounce)
        fr.advance_sample()
        orc.stage("advance")
    st = fr.stats()
    assert st.log_fallbacks == 0 and st.log_inline_entries == 0      # the stages re-allocated the frame in the full layout
    assert np.array_equal(fr.radiance()[..., :3], orc.radiance()[..., :3])
    fr.close()
    # a pool that cannot hold the batch: rt_integrate falls back (and says so), the stages that follow are exact all the same
    fr = capi.Frame(ctx, w, h)
    fr.set_camera(cam)
    fr.set_max_bounces(bounces)
    fr.set_option(capi.OPT_SAMPLES_IN_FLIGHT, 8)
    fr.set_option(capi.OPT_COMPACT_LOG, 1)
    fr.set_option(capi.OPT_DEBUG_LOG_POOL_DIV, 1000000000)
    spp = 3
    for _ in range(spp):                                             # stages first: the compact allocation of a frame nobody has integrated on yet
        fr.generate_rays()
        for bounce in range(bounces + 1):
            fr.intersect(bounce); fr.shade(bounce); fr.intersect_shadow(bounce)
        fr.advance_sample()
    assert fr.stats().log_inline_entries == 0
    orc2 = _oracle.Oracle(w, h, sc)
    orc2.set_camera(cam)
    orc2.set_max_bounces(bounces)
    orc2.integrate(spp)
    assert np.array_equal(fr.radiance()[..., :3], orc2.radiance()[..., :3])
    fr.close()


# (bit 4 -- shadow records' slots stored likeliest occluder first -- ran on a device for the first time in round 5, call 1: modes (31, 1) and
# (31, 0) passed with the other eight and joined the list, profiles/r05_call01_pytest_adaptive_fold_all_modes.log; bits 0 + 3 + 4 are the library default)
_ADAPTIVE_MODES = [(7, 1), (7, 0), (7, 2), (5, 1), (1, 1), (15, 1), (15, 0), (9, 2), (31, 1), (31, 0)]


@pytest.mark.parametrize("mode,shadow_tree", _ADAPTIVE_MODES)
def test_adaptive_fold_is_adopted_and_changes_no_bit(ctx, mode, shadow_tree):
    """RT_CTX_OPT_ADAPTIVE_FOLD (round 4; on by default as mode 1): the first rt_integrate traces a probe frame, a worker thread folds both
    4-wide trees again for the probe rays' measured box passes, and the records are replaced between two rt_integrate calls -- waiting
    for the worker (bit 1) or whenever it is ready (modes 5 and 1: the frame goes on meanwhile) -- and again when a frame looks at the
    scene from somewhere else.  Whatever the fold and whenever it arrives: the reference's radiance, bit for bit.  shadow_tree 0 / 1 / 2:
    the shadow rays share the closest-hit records, walk the measured choice, or walk the backend's own binary tree (whose fold is then the
    one adapted).  Bit 3 (modes 15, 9): the shadow rays' BINARY tree is rotated for the probe rays' crossings before it is folded
    (tree_rotate.h) -- another topology over the reference's leaves, the same verdicts.  The host half alone: tests/test_adaptive_fold.py."""
    import time
    w, h, bounces = 96, 64, 5
    scene = host.Scene(arrays=S.city_block(40_000))
    scene.add_directional_light((-0.6, -1.5, 3.5), (15.0, 10.0, 5.0))
    scene.set_env_path(os.path.join(ROOT, "assets", "ibl", "CGSkies_0036_free.hdr"))
    scene.build_bvh()
    scene.finalize()
    sc = scene.arrays()
    assert len(sc["nodes"]) >= 8192                                  # mode 1 (the default) adapts trees of this size without being forced to
    ctx.set_adaptive_fold(mode)
    ctx.set_shadow_tree(shadow_tree)
    try:
        ctx.upload_scene(sc)
    finally:
        ctx.set_adaptive_fold(capi.ADAPTIVE_FOLD_DEFAULT)
        ctx.set_shadow_tree(1)
    assert "adaptive fold" not in ctx.tree_report()

    def frame_of(cam, probe, rotated):
        """a frame of `cam`, integrated until the fold of probe number `probe` is in place and then some more; against the oracle"""
        fr = capi.Frame(ctx, w, h)
        fr.set_camera(cam)
        fr.set_max_bounces(bounces)
        fr.integrate(2)                                              # the probe; with bit 1 the new fold is in place when this returns
        tag = "adaptive fold (probe %d)" % probe
        for _ in range(500):
            if tag in ctx.tree_report():
                break
            assert not mode & 2, ctx.tree_report()
            time.sleep(0.02)
            fr.integrate(1)                                          # the frame goes on with the fold it has
        report = ctx.tree_report()
        assert tag in report and "closest-hit" in report and "adopted" in report and report.count("adaptive fold") == 1, report
        assert rotated is None or ("rotated" in report) == rotated, report
        fr.integrate(3)                                              # ... and with the new one
        spp = fr.sample_count()
        orc = _oracle.Oracle(w, h, sc)
        orc.set_camera(cam)
        orc.set_max_bounces(bounces)
        orc.integrate(spp)
        assert np.array_equal(fr.radiance()[..., :3], orc.radiance()[..., :3]), report
        fr.close()
        return spp, orc.radiance()[..., :3]

    cam = T.default_camera(w, h)
    spp, want = frame_of(cam, 1, bool(mode & 8))
    # the same view again: nothing to adapt to
    again = capi.Frame(ctx, w, h)
    again.set_camera(cam)
    again.set_max_bounces(bounces)
    again.integrate(spp)
    assert "adaptive fold (probe 1)" in ctx.tree_report()
    assert np.array_equal(again.radiance()[..., :3], want)
    again.close()
    # from the far end of the block, looking back: the view has left the one the folds were made for
    far = T.default_camera(w, h)
    f = np.array([-0.3, -1.0, -0.05]); f /= np.linalg.norm(f)
    r = np.cross(f, [0.0, 0.0, 1.0]); r /= np.linalg.norm(r)
    u = np.cross(r, f)
    for i, k in enumerate("xyz"):
        far["position"][k] = (18.0, 56.0, 1.5)[i]; far["front"][k] = f[i]; far["up"][k] = u[i]
    frame_of(far, 2, None if mode & 8 else False)       # (whether rotating AGAIN pays for the second view is the worker's business)
    # the next upload starts from the surface-area fold again, and with the option off stays there
    ctx.set_adaptive_fold(0)
    try:
        ctx.upload_scene(sc)
    finally:
        ctx.set_adaptive_fold(capi.ADAPTIVE_FOLD_DEFAULT)
    plain = capi.Frame(ctx, w, h)
    plain.set_camera(cam)
    plain.set_max_bounces(bounces)
    plain.integrate(spp)
    assert "adaptive fold" not in ctx.tree_report()
    assert np.array_equal(plain.radiance()[..., :3], want)
    plain.close()


@pytest.mark.parametrize("world,band", [(2, 8), (3, 4), (8, 8)])
def test_tiling_is_bit_invariant(ctx, golden_scenes, world, band):
    """Image-space sharding (the multi-GPU partition) cannot change a pixel."""
    w, h = 72, 60
    sc = golden_scenes["coverage"]
    cam = T.default_camera(w, h)
    full = render(ctx, sc, w, h, cam, 4, 2).radiance()
    img = np.zeros_like(full)
    seen = np.zeros(h, bool)
    rays = 0
    for r in range(world):
        t = render(ctx, sc, w, h, cam, 4, 2, tile_rank=r, tile_count=world, band_height=band)
        rows = t.global_rows()
        assert not seen[rows].any()
        seen[rows] = True
        img[rows] = t.radiance()
        rays += t.stats().closest_rays
    assert seen.all() and np.array_equal(img, full)
    assert rays == render(ctx, sc, w, h, cam, 4, 2).stats().closest_rays


def test_cpp_host_layer_end_to_end(ctx, golden_scenes, golden_radiance):
    """Scene(OBJ) -> Bvh -> Render -> HIPPathTraceIntegrator::Integrate(), i.e. the
    reference's own call sequence (main.cpp:56-72, render.cpp:38-83,172-204)."""
    name = "cornell_64_b4_s2"
    g = golden_radiance
    scene = host.Scene(os.path.join(ROOT, "assets", "CornellBox.obj"))
    scene.add_directional_light((-0.6, -1.5, 3.5), (15.0, 10.0, 5.0))
    r = host.Render(64, 64, scene)
    r.set_max_bounces(4)
    r.set_camera(g[name + "/camera"])
    r.render_frame()                # staged path: the 15 virtuals, one Integrate()
    r.render_frame()
    assert np.array_equal(r.radiance()[..., :3], g[name + "/radiance"])
    assert np.array_equal(r.resolved()[..., :3], g[name + "/resolved"])   # ResolveRadiance ran inside Integrate()
    assert r.sample_count() == 2
    r.set_max_bounces(4)            # SetMaxBounces -> RequestReset
    r.render_samples(2)             # fused fast path
    assert np.array_equal(r.radiance()[..., :3], g[name + "/radiance"])
    r.enable_white_furnace(True)
    r.render_samples(1)
    assert r.sample_count() == 1    # furnace toggle requested a reset
    with pytest.raises(host.RtError, match="blue-noise sampler tables"):
        r.set_blue_noise(True, table_path="/nonexistent/tables.bin")     # missing asset -> loud


def test_furnace_energy_bound(ctx, golden_scenes):
    """White furnace (the reference's built-in check, render.cpp:157-160): with
    the analytic light switched off, albedo 1 and sky 0.5, no pixel can exceed
    0.5 by more than the BSDF's known energy gain... it must stay finite and the
    diffuse-only Cornell box must converge to <= 0.5."""
    sc = dict(golden_scenes["cornell"])
    lights = sc["lights"].copy()
    lights["radiance"]["x"] = 0; lights["radiance"]["y"] = 0; lights["radiance"]["z"] = 0
    sc["lights"] = lights
    fr = render(ctx, sc, 64, 64, T.default_camera(64, 64), 8, 16, furnace=True)
    img = fr.radiance()[..., :3] / 16.0
    assert np.isfinite(img).all() and img.min() >= 0.0
    assert img.mean() <= 0.5 + 1e-3


def test_errors_are_reported_not_swallowed(golden_scenes):
    c = capi.Context(0)
    fr = capi.Frame(c, 16, 16)
    with pytest.raises(capi.RtError, match="no scene uploaded"):
        fr.integrate(1)
    with pytest.raises(capi.RtError, match="max_bounces"):
        fr.set_max_bounces(1000)
    with pytest.raises(capi.RtError, match="bad device ordinal"):
        capi.Context(99)
    bad = dict(golden_scenes["cornell"])
    tris = bad["triangles"].copy()
    tris["mtl_index"][0] = 1000
    bad["triangles"] = tris
    with pytest.raises(capi.RtError, match="material index"):
        c.upload_scene(bad)
    fr.close()
    c.close()


def test_empty_tile_and_single_pixel(ctx, golden_scenes):
    sc = golden_scenes["cornell"]
    cam = T.default_camera(8, 4)
    t = render(ctx, sc, 8, 4, cam, 2, 1, tile_rank=3, tile_count=4, band_height=8)   # owns no rows
    assert t.local_rows == 0 and t.radiance().shape == (0, 8, 4)
    assert t.stats().closest_rays == 0
    one = render(ctx, sc, 1, 1, T.default_camera(1, 1), 3, 2)
    orc = _oracle.Oracle(1, 1, sc)
    orc.set_camera(T.default_camera(1, 1)); orc.set_max_bounces(3); orc.integrate(2)
    assert np.array_equal(one.radiance()[..., :3], orc.radiance()[..., :3])


def test_deep_bounce_limit(ctx, golden_scenes):
    sc = golden_scenes["coverage"]
    cam = T.default_camera(40, 32)
    fr = render(ctx, sc, 40, 32, cam, 16, 1)          # San-Miguel config depth
    orc = _oracle.Oracle(40, 32, sc)
    orc.set_camera(cam); orc.set_max_bounces(16); orc.integrate(1)
    assert np.array_equal(fr.radiance()[..., :3], orc.radiance()[..., :3])
    assert fr.stats().closest_rays == orc.ray_totals()[0]


@pytest.mark.skipif(not _ref.available(), reason="oracle/_ref/libref.so not shipped")
def test_directly_against_the_reference_kernels(ctx, golden_scenes):
    """The reference's own OpenCL kernels (x86 build) run next to the GPU."""
    w, h = 96, 64
    sc = golden_scenes["coverage"]
    cam = T.default_camera(w, h)
    cam["aperture"] = 0.02
    cam["focus_distance"] = 1.8
    ri = _ref.RefIntegrator(w, h, sc, threads=4)
    ri.set_camera(cam); ri.set_max_bounces(8); ri.integrate(3)
    fr = render(ctx, sc, w, h, cam, 8, 3)
    assert rel_l2(fr.radiance()[..., :3], ri.radiance()[..., :3]) < TOL
    assert np.array_equal(fr.radiance()[..., :3], ri.radiance()[..., :3])
    st = fr.stats()
    assert (st.closest_rays, st.shadow_rays) == ri.ray_totals()


@pytest.mark.parametrize("slots", [2, 3, 8])
@pytest.mark.parametrize("variant", [0, 5, 8, 9, 10, 11, 208, 308, 210, 310, 410, 510, 610, 710])
def test_samples_in_flight_and_kernel_variants_are_bit_invariant(ctx, golden_scenes, slots, variant):
    """RT_OPT_SAMPLES_IN_FLIGHT traces several samples of a pixel concurrently; the
    radiance log replays their contributions in the reference's order, so the sum
    is bit-identical to one-sample-at-a-time -- for every traversal kernel variant
    (incl. the 12-entry LDS stack that spills to HBM)."""
    w, h, b, spp = 64, 48, 6, 7          # 7 is not a multiple of any slots value: partial last batch
    sc = golden_scenes["coverage"]
    cam = T.default_camera(w, h)
    base = render(ctx, sc, w, h, cam, b, spp, slots=1)
    ctx.set_wide_bvh(2 if variant in (11, 310, 510) else 1)           # the wide tree's collapse: two BVH2 levels per record / SAH-optimal (default)
    try:
        ctx.upload_scene(sc)
    finally:
        ctx.set_wide_bvh(1)
    fr = capi.Frame(ctx, w, h)
    fr.set_camera(cam)
    fr.set_max_bounces(b)
    fr.set_option(capi.OPT_SAMPLES_IN_FLIGHT, slots)
    fr.set_option(capi.OPT_TRACE_VARIANT, variant % 100)
    fr.set_option(capi.OPT_SHADE_PARTITION, (variant + slots) & 3)     # k_shade with and without the hits-first partition
    # 208 / 308: k_trace2 with extreme loop thresholds (every lane leaves the node loop at once / nobody until all are done)
    # 10 / 11: k_trace_w4 (4-wide quantized tree); 210 / 310: the same with extreme thresholds; 410 / 510: its grid cut down
    # to the live queue counter (a wave per >= 3 / >= 200 rays per lane: a handful of waves, then the minimum of 8)
    fr.set_option(capi.OPT_TRACE_TUNE, {208: 64 | (64 << 8), 308: 1 | (1 << 8), 210: 64 | (64 << 8), 310: 1 | (1 << 8),
                                        410: 3 << 24, 510: (200 << 24) | (4 << 16),
                                        # 610 / 710: chunk mode (64 rays per wave at a time, static chunks, no refill), whole grid / a few waves
                                        610: 1 << 23, 710: (1 << 23) | (5 << 24) | 1 | (64 << 8)}.get(variant, 0))
    # launches this small would all run in chunk mode (RT_OPT_SMALL_LAUNCH_PATHS, 3 M rays): keep the REFILLING form under test
    # for the explicit variants (610 / 710 force chunk mode through the tune word), the automatic choice (5) takes both
    fr.set_option(capi.OPT_SMALL_LAUNCH_PATHS, 3000000 if variant == 5 and slots != 3 else 0)
    fr.set_option(capi.OPT_COMPACT_LOG, 1 if slots == 8 and variant in (5, 9, 10, 610) else 0)     # the compact log layout takes batches of 8
    fr.integrate(spp)
    assert fr.sample_count() == spp
    assert np.array_equal(fr.radiance(), base.radiance(), equal_nan=True)
    a, bs = fr.stats(), base.stats()
    assert (a.closest_rays, a.shadow_rays) == (bs.closest_rays, bs.shadow_rays)
    orc = _oracle.Oracle(w, h, sc)
    orc.set_camera(cam); orc.set_max_bounces(b); orc.integrate(spp)
    assert np.array_equal(fr.radiance()[..., :3], orc.radiance()[..., :3])
    # growing max_bounces after the fact reallocates the log and still matches
    fr.set_max_bounces(b + 3)
    fr.reset()
    fr.integrate(2)
    orc.set_max_bounces(b + 3); orc.integrate(2)
    assert np.array_equal(fr.radiance()[..., :3], orc.radiance()[..., :3])


def test_generic_buffers_mirror_clcontext_semantics(ctx):
    """rt_buffer_* = cl::Buffer + CLContext::WriteBuffer/ReadBuffer/CopyBuffer (cl_context.cpp:96-113)."""
    import ctypes as C
    lib = ctx.lib
    data = np.arange(1024, dtype=np.uint32)
    a, b = C.c_void_p(), C.c_void_p()
    assert lib.rt_buffer_create(ctx.handle, data.nbytes, data.ctypes.data, C.byref(a)) == 0     # CL_MEM_COPY_HOST_PTR
    assert lib.rt_buffer_create(ctx.handle, data.nbytes, None, C.byref(b)) == 0
    assert lib.rt_buffer_size(a) == data.nbytes and lib.rt_buffer_device_ptr(a)
    assert lib.rt_buffer_copy(a, b, 256 * 4, 0, 512 * 4) == 0
    out = np.zeros(512, np.uint32)
    assert lib.rt_buffer_read(b, 0, out.ctypes.data, out.nbytes) == 0                             # blocking read
    assert np.array_equal(out, data[256:768])
    patch = np.full(16, 7, np.uint32)
    assert lib.rt_buffer_write(b, 64, patch.ctypes.data, patch.nbytes) == 0
    assert lib.rt_buffer_read(b, 0, out.ctypes.data, out.nbytes) == 0
    assert (out[16:32] == 7).all() and out[15] == data[256 + 15] and out[32] == data[256 + 32]
    assert lib.rt_buffer_read(b, data.nbytes, out.ctypes.data, 4) != 0                            # out of range -> error
    assert b"out of range" in lib.rt_last_error(ctx.handle)
    assert lib.rt_buffer_destroy(a) == 0 and lib.rt_buffer_destroy(b) == 0
    assert lib.rt_finish(ctx.handle) == 0


def test_kernel_profile_and_stats_accounting(ctx, golden_scenes):
    """RT_OPT_PROFILE_KERNELS brackets every launch with HIP events; counts add up."""
    sc = golden_scenes["coverage"]
    ctx.upload_scene(sc)
    fr = capi.Frame(ctx, 64, 64)
    fr.set_camera(T.default_camera(64, 64)); fr.set_max_bounces(5)
    fr.set_option(capi.OPT_SAMPLES_IN_FLIGHT, 4)
    fr.set_option(capi.OPT_PROFILE, 1)
    fr.integrate(8)                       # two batches of 4
    p = fr.profile()
    assert p.n_raygen == 2 and p.n_trace_closest == 12 and p.n_shade == 12 and p.n_trace_shadow == 12
    assert p.ms_trace_closest > 0 and p.ms_shade > 0
    st = fr.stats()
    assert st.samples == 8 and st.last_active[0] == 4 * 64 * 64
    orc = _oracle.Oracle(64, 64, sc)
    orc.set_camera(T.default_camera(64, 64)); orc.set_max_bounces(5); orc.integrate(8)
    assert (st.closest_rays, st.shadow_rays) == orc.ray_totals()


class _FrameAdapter:
    """Gives capi.Frame / host.Render the surface of the sequence helper."""
    def __init__(self, obj, kind):
        self.o, self.kind, self.request_reset = obj, kind, False
    def set_max_bounces(self, b): self.o.set_max_bounces(b)
    def set_camera(self, c): self.o.set_camera(c)
    def set_aov(self, a):
        if self.kind == "capi":
            self.o.set_option(capi.OPT_AOV, a); self.request_reset = True      # SetAOV -> RequestReset (:470-483)
        else:
            self.o.set_aov(a)
    def enable_denoiser(self, e):
        if self.kind == "capi":
            self.o.set_option(capi.OPT_DENOISER, int(e)); self.request_reset = True
        else:
            self.o.enable_denoiser(e)
    def integrate(self, n):
        if self.kind == "capi":
            if self.request_reset:          # Integrate(): Reset() happens lazily, with the options of that moment
                self.o.reset(); self.request_reset = False
            self.o.integrate(n)
        elif self.kind == "frames":
            for _ in range(n): self.o.render_frame()                # Integrator::Integrate(), all 15 hooks
        else: self.o.render_samples(n)
    def resolve(self): return self.o.resolve() if self.kind == "capi" else self.o.resolve_now()
    def radiance(self): return self.o.radiance()


@pytest.mark.parametrize("kind", ["capi", "frames", "samples"])
def test_aov_viewer_and_temporal_denoiser(ctx, golden_scenes, golden_radiance, kind):
    """GenerateAOV + TemporalAccumulation + the AOV switch of ResolveRadiance: through the C-ABI
    against the golden sequence of the reference kernels, and through Integrator::Integrate()
    (all 15 hooks) / the fused IntegrateSamples path of the C++ host layer against the oracle."""
    from tests.test_oracle_golden import _aov_denoise_sequence
    g = golden_radiance
    cam = g["aov_denoise/camera"]
    if kind == "capi":
        ctx.upload_scene(golden_scenes["coverage"])
        got = _aov_denoise_sequence(_FrameAdapter(capi.Frame(ctx, 64, 48), "capi"), cam)
        want = g
    else:
        scene = host.Scene(os.path.join(ROOT, "assets", "CornellBox.obj"))
        scene.add_directional_light((-0.6, -1.5, 3.5), (15.0, 10.0, 5.0))
        r = host.Render(64, 48, scene)
        got = _aov_denoise_sequence(_FrameAdapter(r, kind), cam)
        want = _aov_denoise_sequence(_oracle.Oracle(64, 48, r.scene_arrays()), cam)
    for k, v in got.items():
        assert np.array_equal(v, want[k], equal_nan=True), k


def test_denoiser_in_the_frame_needs_the_whole_image(ctx, golden_scenes):
    ctx.upload_scene(golden_scenes["cornell"])
    t = capi.Frame(ctx, 32, 32, tile_rank=0, tile_count=2)
    with pytest.raises(capi.RtError, match="whole image"):
        t.set_option(capi.OPT_DENOISER, 1)            # reprojection crosses tile rows: tiles use mode 2 + rt_group_denoise
    t.set_option(capi.OPT_DENOISER, 2)


@pytest.mark.parametrize("tiles", [1, 2, 3])
def test_temporal_denoiser_across_tiles_gather_then_denoise(ctx, golden_scenes, golden_radiance, tiles):
    """The reprojection of TemporalAccumulation crosses tile rows (denoiser.cl:27-79), so with tiling the filter
    runs on the root after ONE gather of radiance + depth + motion vectors (rt_group_denoise).  1, 2 and 3 tiles of
    the golden moving-camera sequence (the reference kernels' own output) -- the tiles live on this box's one GPU
    (rt_group_create_local: device copies instead of RCCL, which wants one device per rank)."""
    g = golden_radiance
    cam = g["aov_denoise/camera"]
    w, h = 64, 48
    ctx.upload_scene(golden_scenes["coverage"])
    frames = [capi.Frame(ctx, w, h, tile_rank=r, tile_count=tiles, band_height=4) for r in range(tiles)]
    grp = capi.Group.create_local(tiles, 0)
    for fr in frames:
        fr.set_max_bounces(3)
        # the golden sequence shows the four AOVs first (4 frames with other cameras' history): replay them so
        # that the frame's prev_camera chain matches, then switch the denoiser on
        for aov in (1, 2, 3, 4):
            fr.set_option(capi.OPT_AOV, aov); fr.set_camera(cam); fr.reset(); fr.integrate(1)
        fr.set_option(capi.OPT_AOV, 0)
        fr.set_option(capi.OPT_DENOISER, 2)          # inputs only: the filter runs on the gathered image
        fr.reset()
    for f in range(5):
        c = cam.copy()
        c["position"]["x"] = 0.02 * f
        for fr in frames:
            fr.set_camera(c)
            fr.integrate(1)
        resolved, radiance = grp.denoise([fr.handle for fr in frames], 0, h, w)
        assert np.array_equal(resolved[..., :3], g["denoise%d/resolved" % f], equal_nan=True), f
        assert np.array_equal(radiance[..., :3], g["denoise%d/radiance" % f], equal_nan=True), f
    grp.close()


def test_aov_viewer_is_tile_invariant(ctx, golden_scenes):
    """The AOVs (albedo, depth, normal, velocity) are per-pixel quantities of the primary hit, so the
    viewer works on tiles: three interleaved tiles reassemble the single-GPU AOV images bit for bit."""
    w, h = 64, 48
    sc = golden_scenes["coverage"]
    cam = T.default_camera(w, h)
    prev = cam.copy(); prev["position"]["x"] = np.float32(0.05)      # a camera move: non-zero velocity AOV
    ctx.upload_scene(sc)
    def frames(**tile):
        fr = capi.Frame(ctx, w, h, **tile)
        fr.set_max_bounces(2)
        fr.set_camera(prev); fr.integrate(1)
        fr.set_camera(cam); fr.reset()
        return fr
    for aov in (1, 2, 3, 4):
        full = frames(); full.set_option(capi.OPT_AOV, aov); full.integrate(1)
        want = full.resolve()
        got = np.zeros_like(want)
        for r in range(3):
            t = frames(tile_rank=r, tile_count=3, band_height=4); t.set_option(capi.OPT_AOV, aov); t.integrate(1)
            got[t.global_rows()] = t.resolve()
        assert np.array_equal(got, want, equal_nan=True), aov
        assert np.abs(want[..., :3]).sum() > 0


@pytest.mark.parametrize("furnace", [False, True])
def test_blue_noise_sampler(ctx, golden_scenes, golden_radiance, furnace):
    """Integrator::SetSamplerType(kBlueNoise): C-ABI path against the reference-kernel golden
    vectors; C++ host path (asset file -> HIPContext::LoadBlueNoiseTables) against the oracle."""
    g = golden_radiance
    sc = golden_scenes["coverage"]
    ctx.upload_scene(sc)
    fr = capi.Frame(ctx, 160, 136)
    with pytest.raises(capi.RtError, match="rt_upload_blue_noise_tables") if not getattr(ctx, "_bn", False) else _nullcontext():
        fr.set_option(capi.OPT_SAMPLER, 1)
    ctx.upload_blue_noise_tables(*S.blue_noise_tables())
    ctx._bn = True
    fr.set_option(capi.OPT_SAMPLER, 1)
    fr.set_option(capi.OPT_WHITE_FURNACE, int(furnace))
    fr.set_camera(g["blue_noise/camera"]); fr.set_max_bounces(9)
    fr.integrate(3)
    assert np.array_equal(fr.radiance()[..., :3], g["blue_noise/radiance_furnace%d" % furnace], equal_nan=True)
    if not furnace:
        scene = host.Scene(os.path.join(ROOT, "assets", "CornellBox.obj"))
        scene.add_directional_light((-0.6, -1.5, 3.5), (15.0, 10.0, 5.0))
        r = host.Render(140, 130, scene)
        r.set_max_bounces(4)
        r.set_blue_noise(True)
        r.set_camera(T.default_camera(140, 130))
        r.render_frame(); r.render_frame()
        orc = _oracle.Oracle(140, 130, r.scene_arrays())
        orc.set_camera(T.default_camera(140, 130)); orc.set_max_bounces(4)
        orc.set_blue_noise(True, S.blue_noise_tables()); orc.integrate(2)
        assert np.array_equal(r.radiance()[..., :3], orc.radiance()[..., :3], equal_nan=True)
        r.set_blue_noise(False)
        r.render_frame()
        assert r.sample_count() == 1          # sampler change requested a reset


class _nullcontext:
    def __enter__(self): return None
    def __exit__(self, *a): return False


def _tiny_scene(env, tris_spec):
    """tris_spec: list of 3x3 vertex arrays; one diffuse + one emissive material."""
    n = len(tris_spec)
    P = np.array(tris_spec, np.float32).reshape(n, 3, 3)
    e1, e2 = P[:, 1] - P[:, 0], P[:, 2] - P[:, 0]
    nrm = np.cross(e1, e2); nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-20)
    N = np.repeat(nrm[:, None, :], 3, axis=1).astype(np.float32)
    U = np.zeros((n, 3, 2), np.float32)
    tris = S.to_triangles([(P, N, U, 0)])
    mats = np.array([S.make_material(kd=(0.8, 0.6, 0.3), ks=(0.4, 0.4, 0.4), roughness=0.3)], dtype=T.packed_material)
    s = host.Scene(arrays=dict(triangles=tris, materials=mats))
    s.add_directional_light((-0.6, -1.5, 3.5), (15.0, 10.0, 5.0))
    s.add_point_light((0.0, 0.0, 3.0), (4.0, 4.0, 4.0))
    s.build_bvh()
    s.set_env_image(env)
    s.finalize()
    return s.arrays()


@pytest.mark.parametrize("variant", [0, 8, 9, 10, 11])
def test_degenerate_bvhs(ctx, env_map, variant):
    """Root-is-a-leaf trees (1 triangle), 2-triangle trees, a leaf with many coincident-centroid
    triangles (the reference's 'all centroids equal' leaf, bvh.cpp:112-123), and a long chain of
    nested triangles that makes a deep, one-sided tree."""
    big = [[(-1, 1, 0), (1, 1, 0), (0, 1, 2)]]
    cases = {
        "one": big,
        "two": big + [[(-1, 2, 0), (1, 2, 0), (0, 2, 2)]],
        # 9 coincident triangles (same centroid) + 1 apart -> one 9-primitive leaf
        "coincident": [[(-1 + 0.0, 1.5, 0), (1, 1.5, 0), (0, 1.5, 2)]] * 9 + big,
        # 70 slabs stacked in depth: every ray pierces many boxes
        "stack": [[(-2, 1 + 0.05 * k, -1), (2, 1 + 0.05 * k, -1), (0, 1 + 0.05 * k, 3)] for k in range(70)],
    }
    for name, spec in cases.items():
        sc = _tiny_scene(env_map, spec)
        cam = T.default_camera(48, 40)
        ctx.upload_scene(sc)
        fr = capi.Frame(ctx, 48, 40)
        fr.set_camera(cam); fr.set_max_bounces(3); fr.set_option(capi.OPT_TRACE_VARIANT, variant)
        fr.set_option(capi.OPT_SMALL_LAUNCH_PATHS, 0 if variant == 11 else 3000000)      # 11: the refilling form, 10: chunk mode (launches this small)
        fr.integrate(3)
        orc = _oracle.Oracle(48, 40, sc)
        orc.set_camera(cam); orc.set_max_bounces(3); orc.integrate(3)
        assert np.array_equal(fr.radiance()[..., :3], orc.radiance()[..., :3], equal_nan=True), (name, variant)
        st = fr.stats()
        assert (st.closest_rays, st.shadow_rays) == orc.ray_totals(), name


def test_axis_aligned_rays_and_select_form_slab_test(ctx, env_map, golden_scenes):
    """k_trace runs the slab test on v_min/v_max_f32 except for rays whose 1/dir has a
    non-finite component (0 * inf = NaN is where minNum/maxNum and the reference's
    compare+select forms differ) -- those keep the select forms.  An overhead light along
    +z makes EVERY shadow ray such a ray (dir = (0, 0, 1), 1/dir = (inf, inf, 1)), over
    axis-aligned geometry whose box planes the ray origins sit on; RT_OPT_TRACE_SELECT_FORM_BOX
    forces the select forms for all rays and must not change a bit."""
    quads = []
    for z, half in ((0.0, 3.0), (0.5, 0.5), (1.0, 0.25)):           # stacked axis-aligned plates
        a, b, c, d = (-half, 1 - half, z), (half, 1 - half, z), (half, 1 + half, z), (-half, 1 + half, z)
        quads += [[a, b, c], [a, c, d]]
    n = len(quads)
    P = np.array(quads, np.float32).reshape(n, 3, 3)
    N = np.tile(np.array([0, 0, 1], np.float32), (n, 3, 1))
    tris = S.to_triangles([(P, N, np.zeros((n, 3, 2), np.float32), 0)])
    mats = np.array([S.make_material(kd=(0.7, 0.7, 0.7), ks=(0.3, 0.3, 0.3), roughness=0.4)], dtype=T.packed_material)
    s = host.Scene(arrays=dict(triangles=tris, materials=mats))
    s.add_directional_light((0.0, 0.0, 1.0), (6.0, 6.0, 6.0))
    s.build_bvh(); s.set_env_image(env_map); s.finalize()
    w, h = 64, 48
    cam = T.default_camera(w, h)
    for sc, bounces in ((s.arrays(), 4), (golden_scenes["coverage"], 6)):
        imgs = []
        # select 2: k_trace_w4, which hands exactly these rays to the BVH2 kernel through its slow list
        for select in (0, 1, 2):
            ctx.upload_scene(sc)
            fr = capi.Frame(ctx, w, h)
            fr.set_camera(cam); fr.set_max_bounces(bounces)
            fr.set_option(capi.OPT_SELECT_FORM_BOX, select & 1)
            fr.set_option(capi.OPT_TRACE_VARIANT, 10 if select == 2 else 5)
            fr.integrate(4)
            imgs.append(fr.radiance()[..., :3].copy())
            st = fr.stats()
            if sc is not golden_scenes["coverage"]:
                # the overhead light makes every shadow ray "slow": the automatic choice sees that at upload and traces the
                # shadow queue with k_trace2 (nothing handed over); k_trace_w4 asked for explicitly hands every one of them over
                assert (st.slow_rays >= st.shadow_rays > 0) if select == 2 else (st.slow_rays < st.shadow_rays // 4)
        orc = _oracle.Oracle(w, h, sc)
        orc.set_camera(cam); orc.set_max_bounces(bounces); orc.integrate(4)
        assert np.array_equal(imgs[0], imgs[1], equal_nan=True)
        assert np.array_equal(imgs[0], imgs[2], equal_nan=True)
        assert np.array_equal(imgs[0], orc.radiance()[..., :3], equal_nan=True)
        assert (st.closest_rays, st.shadow_rays) == orc.ray_totals()


def test_reserve_samples_and_lazy_path_buffers(ctx, golden_scenes):
    """Per-path buffers follow the largest batch requested (rt_frame_reserve_samples or
    rt_integrate) instead of the cap; growing them between batches keeps the sum exact."""
    w, h, b = 64, 48, 5
    sc = golden_scenes["coverage"]
    cam = T.default_camera(w, h)
    base = render(ctx, sc, w, h, cam, b, 9, slots=1)
    ctx.upload_scene(sc)
    fr = capi.Frame(ctx, w, h)
    fr.set_camera(cam); fr.set_max_bounces(b)
    assert fr.reserve_samples(4) == 4
    fr.integrate(2)                      # fits the reservation
    assert fr.reserve_samples(1) == 4    # never shrinks
    fr.integrate(7)                      # grows to 7 in flight with 2 samples already accumulated
    assert fr.reserve_samples(0) == 7
    assert fr.sample_count() == 9
    assert np.array_equal(fr.radiance(), base.radiance(), equal_nan=True)
    assert fr.reserve_samples(100000) == 1024         # clamped to the auto cap at this tile size
    fr.set_option(capi.OPT_SAMPLES_IN_FLIGHT, 3)      # explicit: allocated at once, and the new cap
    assert fr.reserve_samples(50) == 3


def test_rt_render_cli(tmp_path):
    """The headless CLI with the reference's flags (main.cpp:42-53) renders the Cornell box."""
    import subprocess
    exe = os.path.join(ROOT, "raytracing_amd", "rt_render")
    out = tmp_path / "img.pfm"
    r = subprocess.run([exe, "-w", "64", "-h", "48", "--scene", "assets/CornellBox.obj", "--spp", "4", "--bounces", "4",
                        "--out", str(out)], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "Mrays/s" in r.stdout and "gfx950" in r.stdout
    raw = open(out, "rb").read()
    assert raw.startswith(b"PF\n64 48\n-1.0\n")
    img = np.frombuffer(raw[len(b"PF\n64 48\n-1.0\n"):], np.float32).reshape(48, 64, 3)[::-1]   # PFM: bottom row first
    scene = host.Scene(os.path.join(ROOT, "assets", "CornellBox.obj"))
    scene.add_directional_light((-0.6, -1.5, 3.5), (15.0, 10.0, 5.0))
    scene.build_bvh(); scene.finalize()
    orc = _oracle.Oracle(64, 48, scene.arrays())
    orc.set_camera(T.default_camera(64, 48)); orc.set_max_bounces(4); orc.integrate(4)
    assert np.array_equal(img, orc.radiance()[..., :3] / np.float32(4.0))
    bad = subprocess.run([exe, "--scene", "assets/does_not_exist.obj"], cwd=ROOT, capture_output=True, text=True, timeout=60)
    assert bad.returncode == 1 and "Caught exception: Failed to load the scene!" in bad.stderr
    # binary scene cache: written by one run, rendered by the next without parsing or building
    cache, out2 = tmp_path / "cornell.rtscene", tmp_path / "img2.pfm"
    w = subprocess.run([exe, "-w", "64", "-h", "48", "--scene", "assets/CornellBox.obj", "--spp", "1", "--bounces", "4",
                        "--save-cache", str(cache)], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert w.returncode == 0 and cache.exists(), w.stderr
    r2 = subprocess.run([exe, "-w", "64", "-h", "48", "--scene", str(cache), "--spp", "4", "--bounces", "4",
                         "--out", str(out2)], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0, r2.stderr
    assert open(out2, "rb").read() == raw
    # --frames: the reference's own loop (n x Render::RenderFrame(), resolve + Finish() every frame) timed from C++
    fr = subprocess.run([exe, "-w", "64", "-h", "48", "--scene", "assets/CornellBox.obj", "--bounces", "4", "--frames", "5"],
                        cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert fr.returncode == 0 and "5 frames (one Integrate() each" in fr.stdout and "ms per frame" in fr.stdout, fr.stdout + fr.stderr
    # the multi-GPU path of the C++ host (TiledRender: device group + the one RCCL gather), on the GPUs there are
    out3 = tmp_path / "img3.pfm"
    r3 = subprocess.run([exe, "-w", "64", "-h", "48", "--scene", "assets/CornellBox.obj", "--spp", "4", "--bounces", "4",
                         "--gpus", "1", "--tiled", "1", "--out", str(out3)], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r3.returncode == 0, r3.stderr
    assert "on 1 GPUs" in r3.stdout and "gather" in r3.stdout
    assert "RCCL ncclGather, communicator of 1 ranks" in r3.stdout          # ncclCommCount, not the caller's own argument
    assert open(out3, "rb").read() == raw
    # ... and with three tiles: scene built once, one context + integrator + host thread per tile, band assembly -- all three
    # on this box's one GPU (device copies instead of RCCL, which refuses several ranks per device)
    out4 = tmp_path / "img4.pfm"
    r4 = subprocess.run([exe, "-w", "64", "-h", "48", "--scene", "assets/CornellBox.obj", "--spp", "4", "--bounces", "4",
                         "--gpus", "3", "--shared_device", "1", "--out", str(out4)], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r4.returncode == 0, r4.stderr
    assert "on 3 GPUs" in r4.stdout and "device copies on one GPU (local group)" in r4.stdout and "tile 2 on device 0" in r4.stdout
    assert open(out4, "rb").read() == raw


def test_device_group_gather_through_the_c_abi(ctx, golden_scenes):
    """rt_group_*: the one RCCL gather below the C-ABI.  This box has one GPU, so the group has one rank
    (RCCL refuses two ranks per device): communicator set-up, ncclGather, band re-assembly and read-back all run;
    both ways of forming a group (all ranks in this process / one process per rank with a unique id)."""
    w, h, b, spp = 72, 50, 4, 3           # 50 rows: the last band is short
    sc = golden_scenes["coverage"]
    cam = T.default_camera(w, h)
    orc = _oracle.Oracle(w, h, sc)
    orc.set_camera(cam); orc.set_max_bounces(b); orc.integrate(spp)
    ctx.upload_scene(sc)
    fr = capi.Frame(ctx, w, h)
    fr.set_camera(cam); fr.set_max_bounces(b); fr.integrate(spp)
    for make in (lambda: capi.Group.create([0]), lambda: capi.Group.join(1, 0, capi.Group.unique_id(), 0)):
        g = make()
        assert g.size() == 1 and g.local_ranks() == [0]
        assert g.comm_count() == (1, 0)                   # ncclCommCount / ncclCommUserRank: RCCL's own word, not the caller's argument
        img = g.gather_radiance([fr.handle], 0, h, w)
        assert np.array_equal(img, fr.radiance(), equal_nan=True)
        assert np.array_equal(img[..., :3], orc.radiance()[..., :3], equal_nan=True)
        assert g.device_image
        # a frame that is not this rank's tile is refused
        other = capi.Frame(ctx, w, h, tile_rank=0, tile_count=2)
        with pytest.raises(capi.RtError, match="not the tile of this rank"):
            g.gather_radiance([other.handle], 0, h, w)
        with pytest.raises(capi.RtError, match="bad root"):
            g.gather_radiance([fr.handle], 1, h, w)
        g.close()
    # three tiles on this one GPU (local transport): padding of the short last tile, band re-assembly, any root
    tiles = [capi.Frame(ctx, w, h, tile_rank=r, tile_count=3, band_height=4) for r in range(3)]
    for t in tiles:
        t.set_camera(cam); t.set_max_bounces(b); t.integrate(spp)
    g = capi.Group.create_local(3, 0)
    assert g.comm_count(1) == (0, -1)                     # a local group has no RCCL communicator
    for root in (0, 2):
        img = g.gather_radiance([t.handle for t in tiles], root, h, w)
        assert np.array_equal(img, fr.radiance(), equal_nan=True)
    g.close()
    with pytest.raises(capi.RtError, match="one rank per device"):
        capi.Group.create([0, 0])
    # The same list past this library's own check: it reaches ncclCommInitAll -- the in-process path `rt_render --gpus N` and
    # TiledRender use -- and the refusal that comes back is RCCL'S (an ncclResult string, not this library's sentence): the
    # wiring up to communicator creation, proven with the one GPU there is.
    with pytest.raises(capi.RtError) as e:
        capi.Group.create([0, 0], unchecked=True)
    assert "ncclCommInitAll" in str(e.value) and "one rank per device" not in str(e.value), str(e.value)
    g = capi.Group.create([0], unchecked=True)             # ... and one rank still forms a communicator through that entry point
    assert g.comm_count() == (1, 0)
    g.close()
    with pytest.raises(capi.RtError, match="bad device ordinal"):
        capi.Group.create([0, 7])


def test_presented_frames_equal_resolved_frames(ctx, golden_scenes):
    """rt_frame_present (round 4): ResolveRadiance + Finish() as the reference has them -- the frame's kernels have finished when it
    returns, the image travels to the host meanwhile, double-buffered on the device.  Frame by frame it is the image
    rt_frame_resolve returns; presenting again into the same buffer, or resolving while an image is still travelling, is safe."""
    w, h, b = 96, 60, 4
    sc = golden_scenes["coverage"]
    cam = T.default_camera(w, h)
    ctx.upload_scene(sc)
    a, p = capi.Frame(ctx, w, h), capi.Frame(ctx, w, h)
    for f in (a, p):
        f.set_camera(cam); f.set_max_bounces(b)
    buf = np.zeros((h, w, 4), np.float32)
    other = np.zeros((h, w, 4), np.float32)
    for frame_no in range(5):
        a.integrate(1); p.integrate(1)
        want = a.resolve()
        got = p.present(buf if frame_no != 2 else other)    # the third frame into another buffer: copies to different destinations are ordered too
        p.present_wait()
        assert np.array_equal(got, want, equal_nan=True), frame_no
    # back to back without waiting: the copies are ordered, the last one wins
    p.integrate(1); p.present(buf); p.integrate(1); p.present(buf)
    a.integrate(2)
    p.present_wait()
    assert np.array_equal(buf, a.resolve(), equal_nan=True)
    # a synchronous resolve while a presented image is still on its way waits for it first
    p.integrate(1); a.integrate(1)
    p.present(buf)
    assert np.array_equal(p.resolve(), a.resolve(), equal_nan=True)
    a.close(); p.close()


def test_graft_entry_smoke():
    """The driver's round-end smoke check: one small invocation of the hot path against the oracle."""
    import __graft_entry__
    __graft_entry__.smoke()
