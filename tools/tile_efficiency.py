"""Strong-scaling estimate on ONE GPU: rank 0's tile of an N-way split of the bench frame
(interleaved row bands), timed alone.  N ranks render their tiles concurrently on N GPUs,
so job time ~= this tile's time (+ the gather); efficiency = t(1) / (N * t(N))."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from raytracing_amd import capi, host, scenes as S

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=4)
ap.add_argument("--steps", type=int, default=1024, help="samples per pixel of the job (bench default: 8 steps x 128)")
ap.add_argument("--tiles", default="1,2,4,8")
ap.add_argument("--band-height", type=int, default=8)
a = ap.parse_args()
cfg = bench.CONFIGS[a.config]
args = argparse.Namespace(config=a.config, blob_tris=871_200, ball_tris=20_000)
scene, n_tris = bench.build_scene(args, host, S)
base = None
for n in [int(x) for x in a.tiles.split(",")]:
    render = host.Render(cfg["width"], cfg["height"], scene, tile_rank=0, tile_count=n, band_height=a.band_height)
    render.set_camera(host.default_camera(cfg["width"], cfg["height"]))
    render.set_max_bounces(cfg["bounces"])
    render.set_resolve_every_frame(False)
    in_flight = render.reserve_samples(a.steps)
    render.render_samples(min(a.steps, 64)); render.finish()
    frame = host.load().rth_render_frame_handle(render.handle)
    assert capi.load().rt_reset(frame) == 0
    st0 = render.stats()
    t0 = time.perf_counter(); render.render_samples(a.steps); render.finish(); dt = time.perf_counter() - t0
    st = render.stats()
    rays = st.closest_rays + st.shadow_rays - st0.closest_rays - st0.shadow_rays
    if base is None:
        base = dt * n
    print("tiles %d: rank-0 tile %d rows, %d samples in flight, %.1f ms for %d spp, %.0f Mrays/s on this GPU -> x%d = %.0f Mrays/s, efficiency %.3f"
          % (n, render.local_rows, in_flight, dt * 1e3, a.steps, rays / dt / 1e6, n, n * rays / dt / 1e6, base / (n * dt)), flush=True)
    del render
