"""Would TWO frames of the reference's frame-by-frame pattern in flight pay?  (Round 5: every launch of that pattern is its own tail, the machine
idles through the late bounces; RT_OPT_STAGE_PIPES -- the frame's pixels on several streams -- lost, profiles/r05_call02.log.)
A probe with what exists: two Render objects (two contexts, two streams, the scene uploaded twice) take turns, one Integrate() per turn through
the fifteen hooks, nothing waits between turns; against ONE Render doing the same number of frames.  No resolve in either (the question is the
overlap of the traces).  usage: python tools/two_frames_in_flight_probe.py [--config 4] [--frames 96]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from raytracing_amd import capi, host, scenes as S

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=4)
ap.add_argument("--frames", type=int, default=96)
a = ap.parse_args()
cfg = bench.CONFIGS[a.config]
args = argparse.Namespace(config=a.config, scene=None, blob_tris=871_200, ball_tris=20_000, width=cfg["width"], height=cfg["height"], bounces=cfg["bounces"])
scene, n_tris = bench.build_scene(args, host, S)
renders = []
for i in range(2):
    r = host.Render(args.width, args.height, scene)
    r.set_adaptive_fold(27)
    r.set_camera(host.default_camera(args.width, args.height))
    r.set_max_bounces(args.bounces)
    r.set_resolve_every_frame(False)
    r.render_samples(8); r.finish()                      # the fold adaptation happens here
    for _ in range(3):
        r.render_frame()
    r.finish()
    renders.append(r)

def rays(rs):
    return sum(s.closest_rays + s.shadow_rays for s in (r.stats() for r in rs))

out = {}
for name, rs in (("one", renders[:1]), ("two_alternating", renders)):
    r0 = rays(rs)
    t0 = time.perf_counter()
    for i in range(a.frames):
        rs[i % len(rs)].render_frame()
    for r in rs:
        r.finish()
    dt = time.perf_counter() - t0
    out[name] = dict(ms_per_frame=round(dt * 1e3 / a.frames, 4), mrays_per_s=round((rays(rs) - r0) / dt / 1e6, 1))
    print(name, out[name], flush=True)
print(json.dumps(dict(config=a.config, frames=a.frames, resolve=False, **out)))
