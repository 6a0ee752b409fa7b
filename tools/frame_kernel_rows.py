"""Where do k_frame's waves spend a frame?  (RT_OPT_FRAME_KERNEL, raytracing_amd/csrc/frame_kernels.h.)  A few frames of the hooks' pattern with the
frame kernel on, then the per-wave rows of the last launch: rays per wave, 100 MHz ticks in the closest walks / shading / shadow walks, and the
spread of the waves' total times -- the frame lasts as long as its slowest wave.
usage: python tools/frame_kernel_rows.py [--config 4] [--value 1]"""
import argparse, ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from raytracing_amd import capi, host, scenes as S

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=4)
ap.add_argument("--value", type=int, default=1)
a = ap.parse_args()
cfg = bench.CONFIGS[a.config]
args = argparse.Namespace(config=a.config, scene=None, blob_tris=871_200, ball_tris=20_000, width=cfg["width"], height=cfg["height"], bounces=cfg["bounces"])
scene, n_tris = bench.build_scene(args, host, S)
r = host.Render(args.width, args.height, scene)
r.set_adaptive_fold(27)
r.set_camera(host.default_camera(args.width, args.height)); r.set_max_bounces(args.bounces)
r.set_resolve_every_frame(True)
lib = capi.load()
frame = host.load().rth_render_frame_handle(r.handle)
r.render_samples(8); r.finish()
assert lib.rt_set_option(frame, capi.OPT_FRAME_KERNEL, a.value) == 0
for _ in range(4):
    r.render_frame()
r.finish()
t0 = time.perf_counter()
for _ in range(32):
    r.render_frame()
r.finish()
ms = (time.perf_counter() - t0) * 1e3 / 32
n_rows, words = C.c_uint32(), C.c_uint32()
assert lib.rt_frame_debug_frame_rows(frame, None, 0, C.byref(n_rows), C.byref(words)) == 0
rows = np.zeros((n_rows.value, words.value), np.uint32)
assert lib.rt_frame_debug_frame_rows(frame, rows.ctypes.data, n_rows.value, C.byref(n_rows), C.byref(words)) == 0
rays = rows[:, :128].sum(1).astype(np.float64)
tc, ts, th, tt = (rows[:, k].astype(np.float64) / 100.0 for k in (131, 132, 133, 134))       # microseconds
q = lambda x: "min %.0f / median %.0f / p90 %.0f / max %.0f" % (x.min(), np.median(x), np.percentile(x, 90), x.max())
print("config %d, RT_OPT_FRAME_KERNEL = %d: %.3f ms per frame; %d waves" % (a.config, a.value, ms, len(rows)))
print("  rays per wave: " + q(rays))
print("  a wave's total time, us: " + q(tt) + "   (the frame's kernel lasts as long as the slowest)")
print("  ... of which closest walks %.0f %%, shading %.0f %%, shadow walks %.0f %% (means)" % (100 * tc.sum() / tt.sum(), 100 * ts.sum() / tt.sum(), 100 * th.sum() / tt.sum()))
print("  machine utilisation if every wave ended with the slowest: %.2f" % (tt.mean() / tt.max()))
print("  correlation of a wave's time with its rays: %.2f" % np.corrcoef(rays, tt)[0, 1])
