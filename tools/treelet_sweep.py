"""Sweep the BVH record layout (treelet size) on the bench workload; images must be identical."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import capi, host, scenes as S
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tris, mats = S.cornell_blob(871200, 20000)
scene = host.Scene(arrays=dict(triangles=tris, materials=mats))
scene.add_directional_light((-0.6, -1.5, 3.5), (15., 10., 5.))
scene.set_env_path(os.path.join(ROOT, "assets", "ibl", "CGSkies_0036_free.hdr"))
scene.build_bvh(); scene.finalize()
arr = scene.arrays()
ctx = capi.Context(0)
ref = None
for t in [int(x) for x in sys.argv[1].split(",")]:
    ctx.set_treelet_nodes(t)
    ctx.upload_scene(arr)
    fr = capi.Frame(ctx, 1280, 720); fr.set_camera(host.default_camera(1280, 720)); fr.set_max_bounces(8)
    fr.set_option(capi.OPT_PROFILE, 1)
    fr.integrate(32); ctx.finish(); fr.profile(); fr.reset()
    t0 = time.perf_counter(); fr.integrate(64); ctx.finish(); dt = time.perf_counter() - t0
    p = fr.profile(); st = fr.stats(); img = fr.radiance()
    same = True if ref is None else np.array_equal(img, ref, equal_nan=True)
    ref = img if ref is None else ref
    print("treelet %4d: %.3f ms/spp %.0f Mrays/s | closest %.3f shadow %.3f shade %.3f | identical=%s" % (
        t, dt * 1e3 / 64, (st.closest_rays + st.shadow_rays) / dt / 1e6, p.ms_trace_closest / 64, p.ms_trace_shadow / 64,
        p.ms_shade / 64, same), flush=True)
    fr.close()
