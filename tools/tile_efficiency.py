"""Strong-scaling estimate on ONE GPU: EVERY rank's tile of an N-way split of the bench frame
(interleaved row bands) is rendered alone, one after the other, and the job time is the SLOWEST
tile's time.  efficiency = t(1) / (N * t_job(N)).  On N GPUs the tiles render concurrently; what
this cannot show is the gather (rt_group_gather_radiance: 16 B per pixel, 4.1 MB per rank at
1080p -- tens of microseconds at 153 GB/s per xGMI link; bench.py --gpus N times the real one)."""
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
lib = capi.load()
base = None
for n in [int(x) for x in a.tiles.split(",")]:
    times, rays_total, in_flight = [], 0, 0
    for rank in range(n):
        render = host.Render(cfg["width"], cfg["height"], scene, tile_rank=rank, tile_count=n, band_height=a.band_height)
        render.set_camera(host.default_camera(cfg["width"], cfg["height"]))
        render.set_max_bounces(cfg["bounces"])
        render.set_resolve_every_frame(False)
        in_flight = render.reserve_samples(a.steps)
        render.render_samples(min(a.steps, 64)); render.finish()
        frame = host.load().rth_render_frame_handle(render.handle)
        assert lib.rt_reset(frame) == 0
        st0 = render.stats()
        t0 = time.perf_counter(); render.render_samples(a.steps); render.finish(); dt = time.perf_counter() - t0
        st = render.stats()
        rays_total += st.closest_rays + st.shadow_rays - st0.closest_rays - st0.shadow_rays
        times.append(dt)
        del render
    t_job = max(times)
    if base is None:
        base = t_job * n
    print("tiles %d: %d samples in flight, per-rank ms min %.1f / max %.1f (slowest rank %d) for %d spp -> job %.1f ms, %.0f Mrays/s, efficiency %.3f"
          % (n, in_flight, min(times) * 1e3, max(times) * 1e3, times.index(max(times)), a.steps, t_job * 1e3, rays_total / t_job / 1e6,
             base / (n * t_job)), flush=True)
