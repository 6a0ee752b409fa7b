"""One rank's tile of an N-way split (or the whole frame, --tiles 1) rendered as BASELINE's job `--repeat` times after a warm-up, for a kernel trace around it:
    rocprofv3 --kernel-trace --stats -- python tools/tile_job_trace.py --tiles 8 --rank 0 --spp 256
Where does the N = 8 tile's job lose its 13 % (tools/tile_efficiency.py)?  Per kernel: (time in the tile's job) x 8 against the whole frame's."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from raytracing_amd import capi, host, scenes as S

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=4)
ap.add_argument("--tiles", type=int, default=8)
ap.add_argument("--rank", type=int, default=0)
ap.add_argument("--spp", type=int, default=256)
ap.add_argument("--repeat", type=int, default=3)
ap.add_argument("--adaptive-fold", type=int, default=27)
a = ap.parse_args()
cfg = bench.CONFIGS[a.config]
args = argparse.Namespace(config=a.config, blob_tris=871_200, ball_tris=20_000, scene=None, width=cfg["width"], height=cfg["height"], bounces=cfg["bounces"])
scene, n_tris = bench.build_scene(args, host, S)
lib = capi.load()
render = host.Render(cfg["width"], cfg["height"], scene, tile_rank=a.rank, tile_count=a.tiles, band_height=8)
if a.adaptive_fold != capi.ADAPTIVE_FOLD_DEFAULT:
    render.set_adaptive_fold(a.adaptive_fold)
render.set_camera(host.default_camera(cfg["width"], cfg["height"]))
render.set_max_bounces(cfg["bounces"])
render.set_resolve_every_frame(False)
frame = host.load().rth_render_frame_handle(render.handle)
in_flight = render.reserve_samples(a.spp)
render.render_samples(min(a.spp, 64)); render.finish()
times = []
for _ in range(a.repeat):
    assert lib.rt_reset(frame) == 0
    render.finish()
    t0 = time.perf_counter(); render.render_samples(a.spp); render.finish(); times.append(time.perf_counter() - t0)
print("tiles %d rank %d: %d spp, %d samples in flight, job ms %s" % (a.tiles, a.rank, a.spp, in_flight, " ".join("%.2f" % (t * 1e3) for t in times)), flush=True)
