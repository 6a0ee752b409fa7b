"""-m gpu: BASELINE.json's configs at FULL size, through size-independent
properties (the oracle needs ~20 s per 720p sample, so exact comparison is done
on crops/tiles): determinism, tile invariance, furnace bound, ray accounting."""
import numpy as np
import pytest
from tests import _oracle
from raytracing_amd import capi, host, scenes as S, types as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def blob_scene(env_map):
    """Stand-in for config 2 (CornellBox_Dragon.obj 1280x720 8-bounce) at a size
    whose BVH builds in seconds."""
    tris, mats = S.cornell_blob(200_000, 20_000)
    s = host.Scene(arrays=dict(triangles=tris, materials=mats))
    s.add_directional_light((-0.6, -1.5, 3.5), (15.0, 10.0, 5.0))
    nodes = s.build_bvh()
    s.set_env_image(env_map)
    s.finalize()
    return s.arrays()


def test_config2_720p_8_bounce_properties(blob_scene):
    w, h, b = 1280, 720, 8
    ctx = capi.Context(0)
    ctx.upload_scene(blob_scene)
    cam = T.default_camera(w, h)
    def run(spp, **tile):
        fr = capi.Frame(ctx, w, h, **tile)
        fr.set_camera(cam); fr.set_max_bounces(b); fr.integrate(spp)
        return fr
    a = run(2)
    img = a.radiance()
    # the reference arithmetic can produce NaN (e.g. the mirror branch's 1/n.o * max(n.o,0)
    # = inf * 0, material.h:79-81 + :230); such pixels must be rare and identical run to run
    bad = ~np.isfinite(img[..., :3]).all(-1)
    assert bad.mean() < 1e-4 and (img[..., :3][~bad] >= 0).all()
    st = a.stats()
    n = w * h
    # counters cover the last BATCH (2 samples traced together by default)
    assert st.last_active[0] == 2 * n and all(st.last_active[i] >= st.last_active[i + 1] for i in range(1, b))
    assert all(st.last_shadow[i] <= st.last_active[i] for i in range(b + 1))
    assert st.closest_rays <= 2 * (b + 1) * n
    # determinism: same seed (sample indices 0,1) -> same bits, independent of atomics order
    assert np.array_equal(run(2).radiance(), img, equal_nan=True)
    # tile invariance at full resolution: 8 GPUs' worth of tiles reassemble the frame
    out = np.zeros_like(img)
    for r in range(8):
        t = run(2, tile_rank=r, tile_count=8, band_height=8)
        out[t.global_rows()] = t.radiance()
    assert np.array_equal(out, img, equal_nan=True)
    # exact check of a window against the oracle rendering the SAME full-frame pixels:
    # the oracle renders the full frame only at low resolution, so compare a low-res frame
    lo = capi.Frame(ctx, 160, 90); lo.set_camera(T.default_camera(160, 90)); lo.set_max_bounces(b); lo.integrate(1)
    orc = _oracle.Oracle(160, 90, blob_scene)
    orc.set_camera(T.default_camera(160, 90)); orc.set_max_bounces(b); orc.integrate(1)
    assert np.array_equal(lo.radiance()[..., :3], orc.radiance()[..., :3], equal_nan=True)
    assert (lo.stats().closest_rays, lo.stats().shadow_rays) == orc.ray_totals()
    ctx.close()


def test_1080p_and_4k_frames_allocate_and_render(blob_scene):
    """Configs 3-5 sizes (1920x1080, 3840x2160, 16 bounces): the per-pixel state
    fits and the schedule runs; linearity in spp of the running sum."""
    ctx = capi.Context(0)
    ctx.upload_scene(blob_scene)
    for (w, h, b) in ((1920, 1080, 8), (3840, 2160, 16)):
        fr = capi.Frame(ctx, w, h)
        fr.set_camera(T.default_camera(w, h)); fr.set_max_bounces(b)
        fr.integrate(1)
        one = fr.radiance()[::8, ::8, :3].copy()
        fr.integrate(1)
        two = fr.radiance()[::8, ::8, :3]
        ok = np.isfinite(two).all(-1) & np.isfinite(one).all(-1)
        assert ok.mean() > 0.9999 and (two[ok] >= one[ok]).all()
        assert fr.stats().last_active[0] == w * h      # one sample per integrate(1) call
        fr.close()
    ctx.close()


def test_headline_frame_128_samples_in_flight_equals_8_in_flight_bit_for_bit(env_map):
    """The production DEPTH at the production SIZE (VERDICT r02 "what's weak" 1): the headline workload -- the full
    2.8 M-triangle stand-in of BASELINE configs[3], 1920x1080, 8 bounces -- rendered as ONE batch of 128 samples in
    flight (265 M paths: 32-bit path ids, the whole radiance log, log_stride > 2^28) and again as 16 batches of 8.
    Batches of 8 are what bench.py's `parity` pins to the reference's own kernels on this very frame (6 in flight), so
    bit-equality here carries that pin to the launch shape the metric is quoted on.  Needs ~110 GB of HBM."""
    from raytracing_amd import scenes
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = host.Scene(arrays=scenes.city_block(2_800_000))
    s.add_directional_light((-0.6, -1.5, 3.5), (15.0, 10.0, 5.0))
    s.set_env_path(os.path.join(root, "assets", "ibl", "CGSkies_0036_free.hdr"))
    w, h, b, spp = 1920, 1080, 8, 128
    r = host.Render(w, h, s)
    r.set_camera(host.default_camera(w, h))
    r.set_max_bounces(b)
    lib = capi.load()
    frame = host.load().rth_render_frame_handle(r.handle)
    if r.reserve_samples(spp) < spp:
        pytest.skip("this GPU cannot keep 128 samples of a 1080p frame in flight")
    r.render_samples(spp)
    big = r.radiance().copy()
    st_big = r.stats()
    assert st_big.samples_in_flight == spp and st_big.last_active[0] == w * h * spp and st_big.chunk_pixels == w * h
    assert lib.rt_set_option(frame, capi.OPT_SAMPLES_IN_FLIGHT, 8) == 0
    assert lib.rt_reset(frame) == 0
    r.render_samples(spp)
    small = r.radiance()
    st_small = r.stats()
    assert st_small.samples_in_flight == 8 and st_small.last_active[0] == w * h * 8
    diff = ~((big == small) | (np.isnan(big) & np.isnan(small))).all(-1)
    assert not diff.any(), "%d of %d pixels differ, first at %s" % (diff.sum(), diff.size, np.argwhere(diff)[:3].tolist())
    assert (st_small.closest_rays, st_small.shadow_rays) == (st_big.closest_rays, st_big.shadow_rays)     # rt_reset rewinds the counters
    r.close()


def test_a_job_that_does_not_reserve_grows_its_buffers_lean(blob_scene):
    """Round 6: mapping the per-path buffers costs seconds per 100 GB, so rt_integrate grows them with the samples it has been asked for (an eighth of them, at least
    16, never below what 8 GiB hold) unless the caller reserved (rt_frame_reserve_samples) or fixed the count (RT_OPT_SAMPLES_IN_FLIGHT) -- same bits either way."""
    w, h, b = 960, 540, 4
    ctx = capi.Context(0)
    ctx.upload_scene(blob_scene)
    cam = T.default_camera(w, h)
    lean, full = capi.Frame(ctx, w, h), capi.Frame(ctx, w, h)
    for f in (lean, full):
        f.set_camera(cam); f.set_max_bounces(b)
    assert full.reserve_samples(256) >= 128
    lean.integrate(256); full.integrate(256)
    sl, sf = lean.stats(), full.stats()
    assert sl.samples_in_flight < sf.samples_in_flight and sl.path_state_bytes <= 9 * 2**30, (sl.samples_in_flight, sl.path_state_bytes)
    assert np.array_equal(lean.radiance(), full.radiance(), equal_nan=True)
    lean.integrate(2048 - 256)                           # a job that keeps asking reaches the full batch
    assert lean.stats().samples_in_flight >= min(sf.samples_in_flight, 256)
    lean.close(); full.close(); ctx.close()
