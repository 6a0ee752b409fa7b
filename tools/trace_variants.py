"""A/B harness: per-kernel HIP-event times of the bench workload for each trace
kernel variant (RT_OPT_TRACE_VARIANT); checks the images are identical."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import capi, host, scenes as S

ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="0,1,2,3,4")
ap.add_argument("--spp", type=int, default=8)
ap.add_argument("--tris", type=int, default=871200)
ap.add_argument("--width", type=int, default=1280)
ap.add_argument("--height", type=int, default=720)
ap.add_argument("--bounces", type=int, default=8)
ap.add_argument("--waves", default="0")
ap.add_argument("--slots", default="1")
ap.add_argument("--select", default="0", help="RT_OPT_TRACE_SELECT_FORM_BOX values to sweep")
ap.add_argument("--tune", default="0", help="RT_OPT_TRACE_TUNE values to sweep for variants 8-11: node_q[:leaf_q[:rays per hand-out]]")
ap.add_argument("--config", type=int, default=0, help="bench.py config (scene, frame, bounces) instead of --tris/--width/--height/--bounces")
args = ap.parse_args()

if args.config:
    import bench
    cfg = bench.CONFIGS[args.config]
    args.width, args.height, args.bounces = cfg["width"], cfg["height"], cfg["bounces"]
    scene, _ = bench.build_scene(argparse.Namespace(config=args.config, blob_tris=871200, ball_tris=20000), host, S)
else:
    tris, mats = S.cornell_blob(args.tris, 20000)
    scene = host.Scene(arrays=dict(triangles=tris, materials=mats))
    scene.add_directional_light((-0.6, -1.5, 3.5), (15., 10., 5.))
    scene.set_env_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets", "ibl", "CGSkies_0036_free.hdr"))
render = host.Render(args.width, args.height, scene)
render.set_camera(host.default_camera(args.width, args.height))
render.set_max_bounces(args.bounces)
render.set_resolve_every_frame(False)
frame = host.load().rth_render_frame_handle(render.handle)
lib = capi.load()
lib.rt_set_option(frame, capi.OPT_PROFILE, 1)
ref_img = None
def parse_tune(t):
    if t.lower().startswith("0x"):
        return int(t, 16)                      # the raw RT_OPT_TRACE_TUNE word (bit 23 = chunk mode, bits 24..31 = rays per lane)
    a = t.split(":")
    return int(a[0]) | ((int(a[1]) if len(a) > 1 else 0) << 8) | (((int(a[2]) // 16) if len(a) > 2 else 0) << 16)


for v, wv, sl, sel, tune in [(int(x), int(w), int(z), int(q), parse_tune(t)) for x in args.variants.split(",") for w in args.waves.split(",")
                             for z in args.slots.split(",") for q in args.select.split(",")
                             for t in (args.tune.split(",") if int(x) >= 8 else ["0"])]:
    assert lib.rt_set_option(frame, capi.OPT_TRACE_TUNE, tune) == 0
    assert lib.rt_set_option(frame, capi.OPT_SMALL_LAUNCH_PATHS, 0) == 0       # the variant asked for, whatever the launch size
    assert lib.rt_set_option(frame, capi.OPT_SELECT_FORM_BOX, sel) == 0
    assert lib.rt_set_option(frame, capi.OPT_SAMPLES_IN_FLIGHT, sl) == 0
    assert lib.rt_set_option(frame, capi.OPT_TRACE_VARIANT, v) == 0
    assert lib.rt_set_option(frame, capi.OPT_TRACE_WAVES, wv) == 0
    render.set_max_bounces(args.bounces)      # requests a reset
    render.render_samples(max(2, sl)); render.finish()
    prof = capi.rt_profile(); lib.rt_frame_get_profile(frame, prof)
    render.set_max_bounces(args.bounces)
    st0 = render.stats()
    spp = max(args.spp, 2 * sl)
    t = time.perf_counter(); render.render_samples(spp); render.finish(); dt = time.perf_counter() - t
    lib.rt_frame_get_profile(frame, prof)
    st = render.stats()
    rays = st.closest_rays + st.shadow_rays
    img = render.radiance()
    same = True if ref_img is None else np.array_equal(img, ref_img, equal_nan=True)
    if ref_img is None:
        ref_img = img
    print("variant %d tune 0x%08x waves %d slots %d select %d: %.2f ms/spp  %.1f Mrays/s | closest %.3f ms  shadow %.3f ms  shade %.3f ms per spp | closest %.0f Mrays/s shadow %.0f Mrays/s | identical=%s"
          % (v, tune, wv, sl, sel, dt * 1e3 / spp, rays / dt / 1e6, prof.ms_trace_closest / spp, prof.ms_trace_shadow / spp,
             prof.ms_shade / spp, st.closest_rays / prof.ms_trace_closest / 1e3, st.shadow_rays / prof.ms_trace_shadow / 1e3, same), flush=True)
