"""-m gpu: RT_OPT_FRAME_KERNEL (round 5) -- ONE sample per pixel through the stage API (the reference's frame-by-frame pattern, Integrator::Integrate
through the fifteen hooks: src/integrator/integrator.cpp:27-59) as ONE launch of k_frame (raytracing_amd/csrc/frame_kernels.h), in which every wave
carries its own pixels through all the bounces.  Everything per path is what the stage kernels do, so the radiance, the resolved image and the ray
counters must equal the golden vectors of the reference build, the oracle and the stage kernels' own output BIT FOR BIT -- including when the recorded
stages have to be replayed with the stage kernels because somebody looks between two of them."""
import os
import numpy as np
import pytest
from tests.conftest import GOLDEN_CASES
from tests import _oracle
from raytracing_amd import capi, host, scenes as S, types as T

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# RT_OPT_FRAME_KERNEL's value: 1 = every block resident, k >= 2 = k chunks of 64 pixels per wave (more blocks than are resident)
KERNEL_VALUE = int(os.environ.get("RT_TEST_FRAME_KERNEL_VALUE", "1"))


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def stage_sample(fr, bounces):
    fr.generate_rays()
    for b in range(bounces + 1):
        fr.intersect(b); fr.shade(b); fr.intersect_shadow(b)
    fr.advance_sample()


def framed(ctx, w, h, cam, bounces, furnace=False, blue=False, kernel=True):
    fr = capi.Frame(ctx, w, h)
    fr.set_camera(cam); fr.set_max_bounces(bounces)
    fr.set_option(capi.OPT_WHITE_FURNACE, int(furnace))
    if blue:
        fr.set_option(capi.OPT_SAMPLER, 1)
    fr.set_option(capi.OPT_FRAME_KERNEL, KERNEL_VALUE if kernel else 0)
    return fr


@pytest.mark.parametrize("case", GOLDEN_CASES, ids=[c[0] for c in GOLDEN_CASES])
def test_frame_kernel_matches_the_reference_golden_vectors(ctx, case, golden_scenes, golden_radiance):
    name, key, w, h, b, spp, furnace = case
    g = golden_radiance
    ctx.upload_scene(golden_scenes[key])
    fr = framed(ctx, w, h, g[name + "/camera"], b, furnace)
    for _ in range(spp):
        stage_sample(fr, b)
    st = fr.stats()
    assert st.frame_kernel_samples == spp, st.frame_kernel_samples            # every sample really went through k_frame
    assert np.array_equal(fr.radiance()[..., :3], g[name + "/radiance"])
    assert np.array_equal(fr.resolve()[..., :3], g[name + "/resolved"])
    assert (st.closest_rays, st.shadow_rays) == tuple(int(x) for x in g[name + "/totals"])
    assert list(st.last_active[: b + 1]) == list(g[name + "/last_active"])
    assert list(st.last_shadow[: b + 1]) == list(g[name + "/last_shadow"])
    assert fr.sample_count() == spp
    fr.close()


@pytest.mark.parametrize("blue", [False, True])
def test_frame_kernel_on_a_city_block_with_an_axis_aligned_light(ctx, blue):
    """40 K triangles, textures, five lights of which one points exactly down an axis (its shadow rays have two zero direction components: the wide
    walk leaves them to the reference's loop on the exact BVH2, inside k_frame too), both samplers; against the oracle and the stage kernels."""
    w, h, b, spp = 160, 96, 5, 3
    scene = host.Scene(arrays=S.city_block(40_000))
    for d, c in (((-0.6, -1.5, 3.5), (15.0, 10.0, 5.0)), ((0.0, 0.0, 1.0), (2.0, 2.0, 3.0)), ((0.4, 0.3, 1.0), (1.0, 1.0, 1.0)),
                 ((-0.2, 0.9, 0.5), (0.5, 1.0, 0.5)), ((0.7, -0.1, 0.6), (1.0, 0.5, 0.5))):
        scene.add_directional_light(d, c)
    scene.set_env_path(os.path.join(ROOT, "assets", "ibl", "CGSkies_0036_free.hdr"))
    scene.build_bvh(); scene.finalize()
    sc = scene.arrays()
    ctx.upload_scene(sc)
    if blue:
        ctx.upload_blue_noise_tables(*S.blue_noise_tables())
    cam = T.default_camera(w, h)
    plain = framed(ctx, w, h, cam, b, blue=blue, kernel=False)
    fr = framed(ctx, w, h, cam, b, blue=blue)
    for _ in range(spp):
        stage_sample(plain, b); stage_sample(fr, b)
    sp, sf = plain.stats(), fr.stats()
    assert sf.frame_kernel_samples == spp and sp.frame_kernel_samples == 0
    assert sf.slow_rays > 0 and sf.slow_rays == sp.slow_rays                   # the axis-aligned light's shadow rays took the exact-BVH2 loop in both
    assert np.array_equal(fr.radiance(), plain.radiance(), equal_nan=True)
    assert (sf.closest_rays, sf.shadow_rays) == (sp.closest_rays, sp.shadow_rays)
    assert list(sf.last_active[: b + 1]) == list(sp.last_active[: b + 1]) and list(sf.last_shadow[: b + 1]) == list(sp.last_shadow[: b + 1])
    if not blue:
        orc = _oracle.Oracle(w, h, sc)
        orc.set_camera(cam); orc.set_max_bounces(b); orc.integrate(spp)
        assert np.array_equal(fr.radiance()[..., :3], orc.radiance()[..., :3], equal_nan=True)
        assert (sf.closest_rays, sf.shadow_rays) == orc.ray_totals()
    fr.close(); plain.close()


def test_recorded_stages_are_replayed_when_somebody_looks_between_them(ctx, golden_scenes):
    """A debug reader, the radiance mid-sample, stages out of the canonical order, a sample that stops early: the recorded stages run with the stage
    kernels after all and the sample goes on there -- same queues, same image; mixing deferred and replayed samples and rt_integrate changes nothing."""
    w, h, b = 72, 48, 4
    sc = golden_scenes["coverage"]
    cam = T.default_camera(w, h)
    ctx.upload_scene(sc)
    plain = framed(ctx, w, h, cam, b, kernel=False)
    fr = framed(ctx, w, h, cam, b)
    # sample 0: a queue read after the first shade
    for f in (plain, fr):
        f.generate_rays(); f.intersect(0); f.shade(0)
    qa, qb = plain.read_queue(0, 1), fr.read_queue(0, 1)
    oa, ob = np.argsort(qa[1], kind="stable"), np.argsort(qb[1], kind="stable")      # (queue ORDER is the blocks' arrival order at an atomic: by pixel)
    for x, y in zip(qa, qb):
        assert np.array_equal(x[oa], y[ob])
    for f in (plain, fr):
        f.intersect_shadow(0)
        for bounce in range(1, b + 1):
            f.intersect(bounce); f.shade(bounce); f.intersect_shadow(bounce)
        f.advance_sample()
    assert fr.stats().frame_kernel_samples == 0
    # sample 1: deferred all the way; sample 2: the radiance read mid-sample; sample 3: through rt_integrate; sample 4: deferred again
    stage_sample(plain, b); stage_sample(fr, b)
    assert fr.stats().frame_kernel_samples == 1
    for f in (plain, fr):
        f.generate_rays()
        for bounce in range(b + 1):
            f.intersect(bounce); f.shade(bounce); f.intersect_shadow(bounce)
            if bounce == 2:
                mid = f.radiance()
        f.advance_sample()
    plain.integrate(1); fr.integrate(1)
    stage_sample(plain, b); stage_sample(fr, b)
    assert fr.stats().frame_kernel_samples == 2
    # a generate without an advance is refused, a reset drops the recorded sample
    fr.generate_rays()
    with pytest.raises(capi.RtError, match="not advanced"):
        fr.generate_rays()
    assert np.array_equal(fr.radiance(), plain.radiance(), equal_nan=True)      # (materialises the recorded generate; the sample stays open)
    fr.reset(); plain.reset()
    stage_sample(plain, b); stage_sample(fr, b)
    assert np.array_equal(fr.radiance(), plain.radiance(), equal_nan=True)
    orc = _oracle.Oracle(w, h, sc)
    orc.set_camera(cam); orc.set_max_bounces(b); orc.integrate(1)
    assert np.array_equal(fr.radiance()[..., :3], orc.radiance()[..., :3], equal_nan=True)
    fr.close(); plain.close()


def test_frame_kernel_at_the_production_frame_through_the_hooks():
    """1920 x 1080, 8 bounces, the bench's city block: Render::RenderFrame() -- Integrate() through the fifteen hooks of HIPPathTraceIntegrator -- with
    RT_OPT_FRAME_KERNEL against the same frames by the stage kernels: bit-identical radiance, equal counters, every frame one k_frame launch."""
    w, h, b, frames = 1920, 1080, 8, 3
    scene = host.Scene(arrays=S.city_block(400_000))
    scene.add_directional_light((-0.6, -1.5, 3.5), (15.0, 10.0, 5.0))
    scene.set_env_path(os.path.join(ROOT, "assets", "ibl", "CGSkies_0036_free.hdr"))
    lib = capi.load()
    images, stats = [], []
    for kernel in (0, 1):
        r = host.Render(w, h, scene)
        r.set_camera(host.default_camera(w, h)); r.set_max_bounces(b)
        frame = host.load().rth_render_frame_handle(r.handle)
        assert lib.rt_set_option(frame, capi.OPT_FRAME_KERNEL, KERNEL_VALUE if kernel else 0) == 0
        r.set_resolve_every_frame(True)
        for _ in range(frames):
            r.render_frame()
        r.finish()
        images.append(r.radiance().copy()); stats.append(r.stats())
        r.close()
    assert stats[1].frame_kernel_samples == frames and stats[0].frame_kernel_samples == 0
    assert np.array_equal(images[0], images[1], equal_nan=True)
    assert (stats[0].closest_rays, stats[0].shadow_rays) == (stats[1].closest_rays, stats[1].shadow_rays)
    assert list(stats[0].last_active[: b + 1]) == list(stats[1].last_active[: b + 1])


def test_the_measured_choice_times_both_ways_and_changes_no_bit(ctx, golden_scenes):
    """RT_OPT_FRAME_KERNEL = 255: frames 0 - 1 of a scene go through the stage kernels and 2 - 3 through k_frame (warm-up), frames 4 - 19 alternate and are
    timed with HIP events, from frame 20 on the faster way stays -- whichever that is, the accumulated image is the stage kernels'; a new scene upload
    measures again."""
    w, h, b, n = 96, 64, 3, 26
    sc = golden_scenes["coverage"]
    cam = T.default_camera(w, h)
    # (a frame during which a fold adaptation is under way is not one of the sixteen timed ones -- round 6, ADVICE r05: the schedule below is the one
    # of a scene whose folds have settled, so this scene's adaptation is waited for: RT_CTX_OPT_ADAPTIVE_FOLD bit 1)
    assert ctx.lib.rt_ctx_set_option(ctx.handle, 4, capi.ADAPTIVE_FOLD_DEFAULT | 2) == 0
    ctx.upload_scene(sc)
    plain = framed(ctx, w, h, cam, b, kernel=False)
    fr = framed(ctx, w, h, cam, b, kernel=False)
    fr.set_option(capi.OPT_FRAME_KERNEL, 255)
    seen = []
    for i in range(n):
        stage_sample(plain, b); stage_sample(fr, b)
        ctx.finish()
        seen.append(fr.stats().frame_kernel_samples)
    assert seen[:4] == [0, 0, 1, 2] and seen[19] == 10, seen                  # frames 2, 3 and the odd ones of 4 .. 19 were k_frame's
    assert seen[-1] in (10, 10 + n - 20), seen                                # ... and from frame 20 on one way or the other
    assert np.array_equal(fr.radiance(), plain.radiance(), equal_nan=True)
    ctx.upload_scene(sc)                                                      # another upload: the choice is made again
    fr.reset(); plain.reset()
    base = fr.stats().frame_kernel_samples
    for i in range(6):
        stage_sample(plain, b); stage_sample(fr, b)
    assert fr.stats().frame_kernel_samples == base + 3                        # frames 2, 3 and 5 of the new measurement
    assert np.array_equal(fr.radiance(), plain.radiance(), equal_nan=True)
    fr.close(); plain.close()
    assert ctx.lib.rt_ctx_set_option(ctx.handle, 4, capi.ADAPTIVE_FOLD_DEFAULT) == 0


def test_frames_during_a_fold_adaptation_are_not_timed(ctx, golden_scenes):
    """with the adaptation asynchronous (the library's default) the measured choice does not time the frames during which the probe runs, the
    worker folds or the new fold is adopted -- they go through the stage kernels -- and the image stays the stage kernels' whatever happens when."""
    w, h, b, n = 96, 64, 3, 40
    sc = golden_scenes["coverage"]
    cam = T.default_camera(w, h)
    ctx.upload_scene(sc)                                                      # (9161 nodes: above the adaptation's threshold)
    plain = framed(ctx, w, h, cam, b, kernel=False)
    fr = framed(ctx, w, h, cam, b, kernel=False)
    fr.set_option(capi.OPT_FRAME_KERNEL, 255)
    for i in range(n):
        stage_sample(plain, b); stage_sample(fr, b)
    assert np.array_equal(fr.radiance(), plain.radiance(), equal_nan=True)
    fr.close(); plain.close()


def test_frame_kernel_on_the_tiles_of_a_multi_gpu_split(ctx, golden_scenes):
    """A tile of an N-way split (interleaved row bands: rt_frame_desc.tile_rank / tile_count / band_height) through k_frame: the assembled image is
    the single frame's, bit for bit (the RNG is keyed by GLOBAL pixel coordinates, the path id by the tile's local pixel)."""
    w, h, b, spp = 96, 72, 4, 2
    sc = golden_scenes["coverage"]
    cam = T.default_camera(w, h)
    ctx.upload_scene(sc)
    full = framed(ctx, w, h, cam, b, kernel=False)
    for _ in range(spp):
        stage_sample(full, b)
    want = full.radiance()
    for world, band in ((2, 8), (3, 4)):
        img = np.zeros_like(want)
        rays = 0
        for r in range(world):
            fr = capi.Frame(ctx, w, h, tile_rank=r, tile_count=world, band_height=band)
            fr.set_camera(cam); fr.set_max_bounces(b)
            fr.set_option(capi.OPT_FRAME_KERNEL, KERNEL_VALUE)
            for _ in range(spp):
                stage_sample(fr, b)
            st = fr.stats()
            assert st.frame_kernel_samples == spp
            img[fr.global_rows()] = fr.radiance()
            rays += st.closest_rays + st.shadow_rays
            fr.close()
        assert np.array_equal(img, want, equal_nan=True), (world, band)
        fs = full.stats()
        assert rays == fs.closest_rays + fs.shadow_rays
    full.close()


def test_a_camera_set_between_recorded_stages_belongs_to_the_next_sample(ctx, golden_scenes):
    """rt_set_camera between rt_generate_rays and rt_advance_sample.  The stage kernels generate a sample's primary rays at rt_generate_rays,
    so a camera set afterwards moves the NEXT sample; a recorded (deferred) sample must do the same -- the recorded stages run with the camera they were
    recorded under."""
    w, h, b = 72, 48, 3
    sc = golden_scenes["coverage"]
    cam, cam2 = T.default_camera(w, h), T.default_camera(w, h)
    cam2["position"]["x"] += np.float32(0.07)
    ctx.upload_scene(sc)
    plain = framed(ctx, w, h, cam, b, kernel=False)
    fr = framed(ctx, w, h, cam, b)
    for f in (plain, fr):
        f.generate_rays()
        f.intersect(0); f.shade(0)
        f.set_camera(cam2)                                   # between two recorded stages
        f.intersect_shadow(0)
        for bounce in range(1, b + 1):
            f.intersect(bounce); f.shade(bounce); f.intersect_shadow(bounce)
        f.advance_sample()
        stage_sample(f, b)                                   # this one sees cam2 in both
    assert np.array_equal(fr.radiance(), plain.radiance(), equal_nan=True)
    assert fr.stats().frame_kernel_samples == 1              # the second sample; the first was replayed by the stage kernels with the old camera
    fr.close(); plain.close()


def test_k_frame_that_cannot_allocate_falls_back_to_the_stage_kernels(ctx, golden_scenes):
    """a failed allocation of k_frame's per-wave buffers must neither leave a half-made grid behind nor lose the recorded sample."""
    w, h, b = 72, 48, 3
    sc = golden_scenes["coverage"]
    cam = T.default_camera(w, h)
    ctx.upload_scene(sc)
    plain = framed(ctx, w, h, cam, b, kernel=False)
    fr = framed(ctx, w, h, cam, b)
    fr.set_option(capi.OPT_DEBUG_ALLOC_LIMIT, 0xFFFFFFFF)    # the hook: k_frame's buffers "do not fit"
    for _ in range(3):
        stage_sample(plain, b); stage_sample(fr, b)
    assert fr.stats().frame_kernel_samples == 0
    assert np.array_equal(fr.radiance(), plain.radiance(), equal_nan=True)
    assert fr.sample_count() == 3
    fr.set_option(capi.OPT_DEBUG_ALLOC_LIMIT, 0)
    fr.set_option(capi.OPT_FRAME_KERNEL, KERNEL_VALUE)       # asked for again, and now it fits
    stage_sample(plain, b); stage_sample(fr, b)
    assert fr.stats().frame_kernel_samples == 1
    assert np.array_equal(fr.radiance(), plain.radiance(), equal_nan=True)
    fr.close(); plain.close()
