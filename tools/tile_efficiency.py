"""Strong-scaling ESTIMATE on ONE GPU (no node with more than one GPU was available to any round): EVERY rank's tile of an N-way
split of the bench frame (interleaved row bands) is rendered alone, one after the other, and the job time is the SLOWEST tile's time.
efficiency = t(1) / (N * t_job(N)).  On N GPUs the tiles render concurrently; what this cannot show is the gather
(rt_group_gather_radiance: 16 B per pixel, 4.1 MB per rank at 1080p -- tens of microseconds at 153 GB/s per xGMI link; bench.py
--gpus N times the real one).  Both job sizes in one pass over the tiles: BASELINE's 256 spp and bench.py's 1024.
Writes --json (bench.py prints it as `scaling_estimate`, "measured": false)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from raytracing_amd import capi, host, scenes as S

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=4)
ap.add_argument("--spp", default="256,1024", help="samples per pixel of the jobs (BASELINE: 256; bench default: 8 steps x 128)")
ap.add_argument("--tiles", default="1,2,4,8")
ap.add_argument("--band-height", type=int, default=8)
ap.add_argument("--adaptive-fold", type=int, default=27)
ap.add_argument("--json", default=None)
ap.add_argument("--pipelines", type=int, default=0, help="RT_OPT_PIPELINES for the tiles' frames (0 = the library's default, 1)")
a = ap.parse_args()
cfg = bench.CONFIGS[a.config]
args = argparse.Namespace(config=a.config, blob_tris=871_200, ball_tris=20_000, scene=None, width=cfg["width"], height=cfg["height"], bounces=cfg["bounces"])
scene, n_tris = bench.build_scene(args, host, S)
lib = capi.load()
jobs = [int(x) for x in a.spp.split(",")]
base = {}
out = dict(what="every rank's tile of an N-way split rendered ALONE on one GPU, one after the other; job time = the slowest tile; "
                "efficiency = t(1) / (N x t_job(N)); the RCCL gather (4.1 MB per rank at 1080p) is not in it", config=a.config,
           adaptive_fold=a.adaptive_fold, jobs={})
for n in [int(x) for x in a.tiles.split(",")]:
    times = {j: [] for j in jobs}
    rays = {j: 0 for j in jobs}
    in_flight = {j: 0 for j in jobs}
    for rank in range(n):
        render = host.Render(cfg["width"], cfg["height"], scene, tile_rank=rank, tile_count=n, band_height=a.band_height)
        if a.adaptive_fold != capi.ADAPTIVE_FOLD_DEFAULT:
            render.set_adaptive_fold(a.adaptive_fold)
        render.set_camera(host.default_camera(cfg["width"], cfg["height"]))
        render.set_max_bounces(cfg["bounces"])
        render.set_resolve_every_frame(False)
        frame = host.load().rth_render_frame_handle(render.handle)
        if a.pipelines:
            assert lib.rt_set_option(frame, capi.OPT_PIPELINES, a.pipelines) == 0
        for j in jobs:
            in_flight[j] = render.reserve_samples(j)
            render.render_samples(min(j, 64)); render.finish()
            assert lib.rt_reset(frame) == 0
            st0 = render.stats()
            t0 = time.perf_counter(); render.render_samples(j); render.finish(); dt = time.perf_counter() - t0
            st = render.stats()
            rays[j] += st.closest_rays + st.shadow_rays - st0.closest_rays - st0.shadow_rays
            times[j].append(dt)
        del render
    for j in jobs:
        t_job = max(times[j])
        base.setdefault(j, t_job * n)
        eff = base[j] / (n * t_job)
        out["jobs"].setdefault(str(j), {})[str(n)] = dict(samples_in_flight=int(in_flight[j]), rank_ms_min=round(min(times[j]) * 1e3, 1), rank_ms_max=round(t_job * 1e3, 1),
                                                         mrays_per_s=round(rays[j] / t_job / 1e6, 0), efficiency=round(eff, 3), speedup=round(eff * n, 2))
        print("%d spp, tiles %d: %d samples in flight, per-rank ms min %.1f / max %.1f (slowest rank %d) -> job %.1f ms, %.0f Mrays/s, efficiency %.3f"
              % (j, n, in_flight[j], min(times[j]) * 1e3, t_job * 1e3, times[j].index(t_job), t_job * 1e3, rays[j] / t_job / 1e6, eff), flush=True)
if a.json:
    json.dump(out, open(a.json, "w"), indent=1)
