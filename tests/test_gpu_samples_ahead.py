"""-m gpu: RT_OPT_SAMPLES_AHEAD (round 6) -- the reference's frame-by-frame pattern (one Integrator::Integrate per frame at one sample per pixel,
src/render.cpp:197, src/integrator/integrator.cpp:27-59) with a standing camera's NEXT samples traced ahead in batches and replayed one sample per
Integrate().  The contract is the reference's: the radiance after EVERY call is the sum of exactly the samples asked for so far, in sample order --
so every frame of a run with the mode must equal the same frame of a run without it BIT FOR BIT (and the golden vectors / the oracle at the end),
through resets, camera changes, option changes, peeks between two stages, rt_integrate in between and a scene uploaded again."""
import os
import numpy as np
import pytest
from tests.conftest import GOLDEN_CASES
from tests import _oracle
from raytracing_amd import capi, host, scenes as S, types as T

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


def framed(ctx, w, h, cam, bounces, ahead, furnace=False, blue=False, tile=None):
    fr = capi.Frame(ctx, w, h) if tile is None else capi.Frame(ctx, w, h, tile_rank=tile[0], tile_count=tile[1], band_height=tile[2])
    fr.set_camera(cam); fr.set_max_bounces(bounces)
    fr.set_option(capi.OPT_WHITE_FURNACE, int(furnace))
    if blue:
        fr.set_option(capi.OPT_SAMPLER, 1)
    fr.set_option(capi.OPT_SAMPLES_AHEAD, ahead)
    return fr


@pytest.mark.parametrize("case", GOLDEN_CASES, ids=[c[0] for c in GOLDEN_CASES])
def test_samples_ahead_matches_the_reference_golden_vectors(ctx, case, golden_scenes, golden_radiance):
    """The golden cases' sample counts, traced through the stage API with batches of up to 4 ahead: radiance and resolved image are the reference build's."""
    name, key, w, h, b, spp, furnace = case
    g = golden_radiance
    ctx.upload_scene(golden_scenes[key])
    fr = framed(ctx, w, h, g[name + "/camera"], b, 4, furnace)
    for _ in range(spp):
        stage_sample(fr, b)
    st = fr.stats()
    assert np.array_equal(fr.radiance()[..., :3], g[name + "/radiance"])
    assert np.array_equal(fr.resolve()[..., :3], g[name + "/resolved"])
    assert fr.sample_count() == spp and st.samples == spp
    if spp > 3:
        assert st.samples_from_banks == spp - 3, (st.samples_from_banks, spp)    # three quiet samples, then every one out of a bank
    fr.close()


@pytest.mark.parametrize("ahead", [2, 4, 1, 256 + 4, 64])
def test_every_frame_equals_the_plain_frame(ctx, golden_scenes, ahead):
    """32 frames, the radiance read after every one of them: with the mode = without it, bit for bit; the totals of fully consumed runs agree."""
    w, h, b, frames = 96, 64, 4, 32
    sc = golden_scenes["coverage"]
    cam = T.default_camera(w, h)
    ctx.upload_scene(sc)
    plain = framed(ctx, w, h, cam, b, 0)
    fr = framed(ctx, w, h, cam, b, ahead)
    for i in range(frames):
        stage_sample(plain, b); stage_sample(fr, b)
        assert np.array_equal(fr.radiance(), plain.radiance(), equal_nan=True), "frame %d" % i
        assert fr.sample_count() == plain.sample_count() == i + 1
    sf, sp = fr.stats(), plain.stats()
    assert sp.samples_from_banks == 0 and sp.samples_ahead == 0
    assert sf.samples_from_banks == frames - 3
    assert sf.samples_ahead >= 1                                                  # something is always traced ahead of a quiet camera ...
    assert sf.closest_rays > sp.closest_rays and sf.shadow_rays > sp.shadow_rays  # ... and its rays are in the totals (rt_stats.samples_ahead says so)
    orc = _oracle.Oracle(w, h, sc)
    orc.set_camera(cam); orc.set_max_bounces(b); orc.integrate(frames)
    assert np.array_equal(fr.radiance()[..., :3], orc.radiance()[..., :3], equal_nan=True)
    assert np.array_equal(fr.resolve(), plain.resolve(), equal_nan=True)
    fr.close(); plain.close()


def test_resets_cameras_options_and_peeks_drop_what_was_traced_ahead(ctx, golden_scenes):
    """Whatever invalidates the speculation -- rt_reset, another camera (with and without a reset), max_bounces, the sampler's furnace switch, a radiance
    read between two stages of a sample that sits in a bank, a queue read, rt_integrate in between -- the sum stays the plain frame's."""
    w, h, b = 80, 56, 3
    sc = golden_scenes["coverage"]
    cam = T.default_camera(w, h)
    cam2 = T.default_camera(w, h)
    cam2["position"]["x"] += np.float32(0.05)
    ctx.upload_scene(sc)
    plain = framed(ctx, w, h, cam, b, 0)
    fr = framed(ctx, w, h, cam, b, 4)

    def both(fn):
        fn(plain); fn(fr)

    def same(what):
        assert np.array_equal(fr.radiance(), plain.radiance(), equal_nan=True), what
        assert fr.sample_count() == plain.sample_count(), what

    for _ in range(7):
        both(lambda f: stage_sample(f, b))
    same("quiet run")
    assert fr.stats().samples_from_banks == 4
    # a reset with batches in flight, then straight on
    both(lambda f: f.reset())
    for _ in range(6):
        both(lambda f: stage_sample(f, b))
    same("after a reset")
    # another camera WITHOUT a reset (the sum goes on with the new view: whatever was traced ahead was traced for the old one)
    both(lambda f: f.set_camera(cam2))
    for _ in range(6):
        both(lambda f: stage_sample(f, b))
    same("after a camera change without reset")
    # ... and back, with one
    both(lambda f: (f.set_camera(cam), f.reset()))
    for _ in range(5):
        both(lambda f: stage_sample(f, b))
    same("camera back")
    banked = fr.stats().samples_from_banks
    assert banked > 4
    # a peek between two stages of a sample that sits in a bank: the frame traces that sample itself after all
    def peek(f):
        f.generate_rays()
        for bounce in range(b + 1):
            f.intersect(bounce); f.shade(bounce); f.intersect_shadow(bounce)
            if bounce == 1:
                f.peeked = f.radiance()                    # (between two bounces: between rt_shade and rt_intersect_shadow the direct samples are still tentative)
        f.advance_sample()
    both(peek)
    assert np.array_equal(fr.peeked, plain.peeked, equal_nan=True)
    same("after a mid-sample read")
    assert fr.stats().samples_from_banks == banked                               # that sample did not come out of a bank
    for _ in range(5):
        both(lambda f: stage_sample(f, b))
    same("speculation resumed")
    assert fr.stats().samples_from_banks > banked
    # rt_integrate between stage samples
    both(lambda f: f.integrate(3))
    for _ in range(5):
        both(lambda f: stage_sample(f, b))
    same("after rt_integrate")
    # fewer bounces (the log's layout changes), furnace on
    both(lambda f: f.set_max_bounces(2))
    both(lambda f: f.reset())
    for _ in range(6):
        both(lambda f: stage_sample(f, 2))
    same("max_bounces 2")
    both(lambda f: (f.set_option(capi.OPT_WHITE_FURNACE, 1), f.reset()))
    for _ in range(6):
        both(lambda f: stage_sample(f, 2))
    same("furnace")
    # a generate without an advance is refused in the mode too
    fr.generate_rays()
    with pytest.raises(capi.RtError, match="not advanced"):
        fr.generate_rays()
    fr.reset(); plain.reset()
    both(lambda f: stage_sample(f, 2))
    same("after the refused call")
    fr.close(); plain.close()


def test_a_scene_uploaded_again_drops_the_banks(ctx, golden_scenes):
    w, h, b = 64, 48, 3
    cam = T.default_camera(w, h)
    ctx.upload_scene(golden_scenes["coverage"])
    fr = framed(ctx, w, h, cam, b, 8)
    for _ in range(6):
        stage_sample(fr, b)
    assert fr.stats().samples_ahead > 0
    key = next(k for k in golden_scenes if k != "coverage")
    ctx.upload_scene(golden_scenes[key])                                          # batches of the old scene are in flight
    fr.reset()
    plain = framed(ctx, w, h, cam, b, 0)
    for _ in range(8):
        stage_sample(fr, b); stage_sample(plain, b)
    assert np.array_equal(fr.radiance(), plain.radiance(), equal_nan=True)
    fr.close(); plain.close()


def test_the_integrator_default_through_the_fifteen_hooks(golden_scenes):
    """host.Render = the C++ Integrator through its fifteen hooks with HIPPathTraceIntegrator's defaults (RT_OPT_SAMPLES_AHEAD = 1, RT_OPT_FRAME_KERNEL
    = 255): 40 x RenderFrame() against RenderSamples(40) of a second Render -- the same sum bit for bit -- with a camera change in the middle."""
    w, h, b, frames = 128, 72, 4, 40
    scene = host.Scene(os.path.join(ROOT, "assets", "CornellBox.obj"))
    scene.add_directional_light((-0.6, -1.5, 3.5), (15.0, 10.0, 5.0))
    a = host.Render(w, h, scene)
    cam = host.default_camera(w, h)
    a.set_camera(cam); a.set_max_bounces(b)
    for _ in range(frames):
        a.render_frame()
    st = a.stats()
    assert st.samples == frames and st.samples_from_banks >= frames - 8, (st.samples, st.samples_from_banks)
    got = a.radiance()
    scene2 = host.Scene(os.path.join(ROOT, "assets", "CornellBox.obj"))
    scene2.add_directional_light((-0.6, -1.5, 3.5), (15.0, 10.0, 5.0))
    r = host.Render(w, h, scene2)
    r.set_camera(cam); r.set_max_bounces(b)
    r.render_samples(frames)
    assert np.array_equal(got, r.radiance(), equal_nan=True)
    # the camera moves: a reset, then a quiet run again
    cam2 = host.default_camera(w, h)
    cam2["position"]["z"] += np.float32(0.1)
    a.set_camera(cam2); r.set_camera(cam2)
    for _ in range(12):
        a.render_frame()
    r.render_samples(12)
    assert a.stats().samples == 12
    assert np.array_equal(a.radiance(), r.radiance(), equal_nan=True)
    a.close(); r.close()


def test_the_same_camera_set_before_every_frame_keeps_the_mode_going(ctx, golden_scenes):
    """Render::RenderFrame sets the camera before every Integrate() (src/render.cpp:188-197): the same camera again is not a change."""
    w, h, b = 64, 48, 3
    cam = T.default_camera(w, h)
    ctx.upload_scene(golden_scenes["coverage"])
    fr, plain = framed(ctx, w, h, cam, b, 4), framed(ctx, w, h, cam, b, 0)
    for _ in range(12):
        for f in (fr, plain):
            f.set_camera(cam)
            stage_sample(f, b)
    assert fr.stats().samples_from_banks == 9
    assert np.array_equal(fr.radiance(), plain.radiance(), equal_nan=True)
    fr.close(); plain.close()


def test_a_tile_of_an_image(ctx, golden_scenes):
    """A frame that owns every second band of the image (rt_frame_desc's tiling): its banks are tiles too."""
    w, h, b = 96, 64, 3
    cam = T.default_camera(w, h)
    ctx.upload_scene(golden_scenes["coverage"])
    tiles = []
    for ahead in (0, 4):
        fr = capi.Frame(ctx, w, h, tile_rank=1, tile_count=2, band_height=8)
        fr.set_camera(cam); fr.set_max_bounces(b); fr.set_option(capi.OPT_SAMPLES_AHEAD, ahead)
        for _ in range(12):
            stage_sample(fr, b)
        tiles.append(fr)
    assert tiles[1].stats().samples_from_banks == 9
    assert np.array_equal(tiles[0].radiance(), tiles[1].radiance(), equal_nan=True)
    for fr in tiles:
        fr.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("RT_SEQ_FIRST", "0")), int(os.environ.get("RT_SEQ_SEEDS", "16"))))   # RT_SEQ_SEEDS=2000 for a campaign
def test_random_sequences_of_calls_equal_the_plain_frame(ctx, golden_scenes, seed):
    """A random walk through the API on three frames at once -- one with batches traced ahead (a random depth, one stream or two), one whose stage samples go through
    the one-launch frame kernel (RT_OPT_FRAME_KERNEL 1 / 2 / 3 / the measured choice), one plain: stage samples,
    rt_integrate of a few samples in between, resets, the same camera set again, another camera with and without a reset, another bounce limit, the other sampler,
    another depth of the mode, the radiance read after every call and sometimes between a sample's stages.  After EVERY call all three hold the same image and the
    same sample count, bit for bit; where one camera has been in place since the last reset, both are the oracle's at the end."""
    rng = np.random.default_rng(77000 + seed)
    key = ("cornell", "coverage")[seed % 2]
    sc = golden_scenes[key]
    w, h = int(rng.integers(16, 72)), int(rng.integers(12, 48))
    bounces = int(rng.integers(1, 6))
    cams = [T.default_camera(w, h)]
    for _ in range(2):
        c = cams[0].copy()
        c["position"]["x"] += np.float32(rng.uniform(-0.3, 0.3)); c["position"]["z"] += np.float32(rng.uniform(-0.2, 0.2))
        cams.append(c)
    if seed % 3 == 1:                                                  # ... and a camera far enough away to leave the view the folds were adapted to
        c = cams[0].copy()
        a = np.float32(0.7)
        for vec in ("front", "up"):
            x, y = float(c[vec]["x"]), float(c[vec]["y"])
            c[vec]["x"], c[vec]["y"] = np.float32(np.cos(a) * x - np.sin(a) * y), np.float32(np.sin(a) * x + np.cos(a) * y)
        cams.append(c)
    depths = (2, 3, 4, 8, 1, 256 + 2, 256 + 5, 64)
    # every third seed: the fold adaptation as the library ships it (asynchronous: probe behind a frame, worker thread, pointer exchange between two calls), on these
    # small trees too (bit 2), rotations and occluder order included, re-armed without a rate limit -- folds change under all three frames while the walk goes on
    adaptive = seed % 3 == 1
    if adaptive:
        ctx.set_adaptive_fold(1 | 4 | 8 | 16)
        assert capi.load().rt_ctx_set_option(ctx.handle, 5, 0) == 0      # RT_CTX_OPT_ADAPT_MIN_INTERVAL_MS
    try:
        ctx.upload_scene(sc)
    finally:
        if adaptive:
            ctx.set_adaptive_fold(capi.ADAPTIVE_FOLD_DEFAULT)
            assert capi.load().rt_ctx_set_option(ctx.handle, 5, 500) == 0
    # every fourth seed: the three frames are one TILE of the image (rank r of 2 - 4, bands of 1 - 8 rows), the multi-GPU path's decomposition
    tile = (int(rng.integers(0, 4)) % (2 + seed % 3), 2 + seed % 3, int(rng.choice([1, 2, 4, 8]))) if seed % 4 == 2 else None
    plain = framed(ctx, w, h, cams[0], bounces, 0, tile=tile)
    fr = framed(ctx, w, h, cams[0], bounces, int(depths[rng.integers(0, len(depths))]), tile=tile)
    fk = framed(ctx, w, h, cams[0], bounces, 0, tile=tile)             # a third frame: no samples ahead, its stage samples through the one-launch frame kernel where eligible
    fk.set_option(capi.OPT_FRAME_KERNEL, int(rng.choice([1, 2, 3, 255])))
    both = (plain, fr, fk)
    denoiser, aov = 0, 0
    since_reset = []                                                   # what the oracle has to repeat at the end: (camera index, bounces, blue, samples) runs since the last reset
    cam_i, blue = 0, False

    def same(what):
        want = plain.radiance()
        shown = plain.resolve() if plain.sample_count() and (tile is None or denoiser == 0) else None
        for name, f in (("samples ahead", fr), ("frame kernel", fk)):
            assert f.sample_count() == plain.sample_count(), (seed, what, name)
            assert np.array_equal(f.radiance(), want, equal_nan=True), (seed, what, name)
            if shown is not None:                                      # what the reference would show: ResolveRadiance (and the AOV / denoiser when switched on)
                assert np.array_equal(f.resolve(), shown, equal_nan=True), (seed, what, name, "resolved")

    oracle_ok = [True]
    def was_reset():                                                   # cl_pt_integrator.cpp:497-508: with the denoiser on, Reset() clears the sum and NOT the sample counter
        since_reset.clear()
        oracle_ok[0] = denoiser == 0

    def note(n):
        if since_reset and since_reset[-1][:3] == (cam_i, bounces, blue):
            since_reset[-1] = (cam_i, bounces, blue, since_reset[-1][3] + n)
        else:
            since_reset.append((cam_i, bounces, blue, n))

    for step in range(int(rng.integers(12, 40))):
        op = int(rng.choice([0, 0, 0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10]))
        if op == 0:                                                    # one frame the reference's way
            peek = rng.random() < 0.15
            for f in both:
                f.generate_rays()
                for b in range(bounces + 1):
                    f.intersect(b); f.shade(b); f.intersect_shadow(b)
                    if peek and b == bounces // 2:
                        f.radiance()                                   # a read between two bounces (after the shadow stage: allowed)
                f.advance_sample()
            note(1)
        elif op == 1:                                                  # a few samples in one call
            k = int(rng.integers(1, 5))
            for f in both:
                f.integrate(k)
            note(k)
        elif op == 2:
            for f in both:
                f.reset()
            was_reset()
        elif op == 3:                                                  # the integrator sets the camera before every frame: the same one changes nothing
            for f in both:
                f.set_camera(cams[cam_i])
        elif op == 4:                                                  # another camera, the reference's way (a reset follows)
            cam_i = int(rng.integers(0, len(cams)))
            for f in both:
                f.set_camera(cams[cam_i]); f.reset()
            was_reset()
        elif op == 5:                                                  # ... and without a reset: the sum goes on with the new camera's samples
            cam_i = int(rng.integers(0, len(cams)))
            for f in both:
                f.set_camera(cams[cam_i])
        elif op == 6:
            bounces = int(rng.integers(1, 6))
            for f in both:
                f.set_max_bounces(bounces); f.reset()
            was_reset()
        elif op == 7:
            blue = not blue
            if blue:
                ctx.upload_blue_noise_tables(*S.blue_noise_tables())
            for f in both:
                f.set_option(capi.OPT_SAMPLER, int(blue)); f.reset()
            was_reset()
        elif op == 8:
            fr.set_option(capi.OPT_SAMPLES_AHEAD, int(depths[rng.integers(0, len(depths))]))
        elif op == 9:                                                  # the temporal denoiser on / off (EnableDenoiser -> RequestReset); a tile only collects its inputs (mode 2)
            denoiser = 0 if denoiser else (1 if tile is None else 2)
            for f in both:
                f.set_option(capi.OPT_DENOISER, denoiser); f.reset()
            was_reset()
        else:                                                          # another AOV (SetAOV -> RequestReset)
            aov = int(rng.integers(0, 5))
            for f in both:
                f.set_option(capi.OPT_AOV, aov); f.reset()
            was_reset()
        if os.environ.get("RT_SEQ_TRACE"):
            print("step", step, "op", op, "count", plain.sample_count(), "noted", since_reset, "denoiser", denoiser, "aov", aov, flush=True)
        same("step %d op %d" % (step, op))
    for _ in range(int(rng.integers(0, 7))):                           # a quiet tail: the mode is (or gets) going when the comparison with the oracle is made
        for f in both:
            stage_sample(f, bounces)
        note(1)
        same("tail")
    if oracle_ok[0] and len(since_reset) == 1 and fr.sample_count() <= 48:             # (one camera since the last reset: the oracle's set_camera starts its sum again, like the reference's)
        orc = _oracle.Oracle(w, h, sc)
        ci, bb, bl, n = since_reset[0]
        orc.set_camera(cams[ci]); orc.set_max_bounces(bb); orc.set_blue_noise(bl, S.blue_noise_tables())
        orc.integrate(n)
        assert orc.sample_count() == fr.sample_count(), seed
        rows = fr.global_rows() if tile is not None else slice(None)
        assert np.array_equal(fr.radiance()[..., :3], orc.radiance()[..., :3][rows], equal_nan=True), seed
    fr.close(); plain.close(); fk.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("RT_HOST_SEQ_FIRST", "0")), int(os.environ.get("RT_HOST_SEQ_SEEDS", "6"))))   # RT_HOST_SEQ_SEEDS=300 for a campaign
def test_random_sequences_through_the_integrator_equal_the_plain_integrator(seed):
    """The same walk one level up, where the reference's own code would call: two host.Render objects -- the C++ Integrator with HIPPathTraceIntegrator behind its fifteen
    hooks -- on the same scene, one on the integrator's defaults (samples ahead: automatic; the frame kernel: measured choice), one with both switched off on its frame:
    RenderFrame(), RenderSamples(k), SetCameraData (the same camera and another: RequestReset), SetMaxBounces, EnableWhiteFurnace, EnableDenoiser, SetAOV in a random
    order; after every call the same radiance, the same resolved image and the same sample count."""
    rng = np.random.default_rng(91000 + seed)
    w, h = int(rng.integers(24, 96)), int(rng.integers(16, 64))
    renders = []
    for _ in range(2):
        scene = host.Scene(os.path.join(ROOT, "assets", "CornellBox.obj"))
        scene.add_directional_light((-0.6, -1.5, 3.5), (15.0, 10.0, 5.0))
        renders.append(host.Render(w, h, scene))
    a, r = renders
    frame_r = host.load().rth_render_frame_handle(r.handle)
    lib = capi.load()
    assert lib.rt_set_option(frame_r, capi.OPT_SAMPLES_AHEAD, 0) == 0 and lib.rt_set_option(frame_r, capi.OPT_FRAME_KERNEL, 0) == 0
    cams = [host.default_camera(w, h)]
    for _ in range(2):
        c = cams[0].copy()
        c["position"]["x"] += np.float32(rng.uniform(-0.2, 0.2)); c["position"]["z"] += np.float32(rng.uniform(-0.2, 0.2))
        cams.append(c)
    bounces = int(rng.integers(1, 6))
    for x in renders:
        x.set_camera(cams[0]); x.set_max_bounces(bounces)
    denoiser, furnace = False, False
    for step in range(int(rng.integers(10, 36))):
        op = int(rng.choice([0, 0, 0, 0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7]))
        for x in renders:
            if op == 0: x.render_frame()
            elif op == 1: x.render_samples(1 + step % 3)
            elif op == 2: x.set_camera(cams[0])
            elif op == 3: x.set_camera(cams[1 + step % 2])
            elif op == 4: x.set_max_bounces(1 + (bounces + step) % 5)
            elif op == 5: x.enable_white_furnace(not furnace)
            elif op == 6: x.enable_denoiser(not denoiser)
            else: x.set_aov(step % 5)
        if op == 5: furnace = not furnace
        if op == 6: denoiser = not denoiser
        for x in renders:
            x.finish()
        assert a.sample_count() == r.sample_count(), (seed, step, op)
        assert np.array_equal(a.radiance(), r.radiance(), equal_nan=True), (seed, step, op)
        if a.sample_count():
            assert np.array_equal(a.resolved(), r.resolved(), equal_nan=True), (seed, step, op, "resolved")
    a.close(); r.close()
