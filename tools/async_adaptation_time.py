"""How long the library's DEFAULT (asynchronous) fold adaptation takes to land while a job renders: batches of 16 samples until the tree report shows the adoption.
    python tools/async_adaptation_time.py [--config 4]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from raytracing_amd import capi, host, scenes as S

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=4)
ap.add_argument("--device-fold", type=int, default=None, help="RT_CTX_OPT_DEVICE_FOLD")
a = ap.parse_args()
cfg = bench.CONFIGS[a.config]
args = argparse.Namespace(config=a.config, blob_tris=871_200, ball_tris=20_000, scene=None)
scene, n_tris = bench.build_scene(args, host, S)
render = host.Render(cfg["width"], cfg["height"], scene, ctx_options=((7, a.device_fold),) if a.device_fold is not None else ())
render.set_camera(host.default_camera(cfg["width"], cfg["height"]))
render.set_max_bounces(cfg["bounces"])
render.set_resolve_every_frame(False)
render.reserve_samples(16)
t0 = time.perf_counter()
batches = 0
while "adaptive fold (probe 1)" not in render.tree_report() and time.perf_counter() - t0 < 60:
    render.render_samples(16 if a.config != 5 else 4); render.finish(); batches += 1
t = time.perf_counter() - t0
line = [l for l in render.tree_report().split("\n") if "adaptive fold" in l]
print("config %d: adopted after %.2f s and %d batches; %s" % (a.config, t, batches, line[0][line[0].find("; ") + 2:][:700] if line else "never"))
