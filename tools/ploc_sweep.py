"""What the device-built shadow tree (PLOC, raytracing_amd/csrc/ploc_kernels.h) costs against own_bvh.h's host-built one, by the metric both minimise (the sum over
the interior boxes of: 50 % isotropic half-area + projected area along the light), over the search radius and the frame of the Morton order.  Needs a GPU.
usage: python tools/ploc_sweep.py [--tris 700000] [--scene city|blob]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import capi, host, scenes as S
from tests.test_own_tree import own_tree

ap = argparse.ArgumentParser()
ap.add_argument("--tris", type=int, default=700_000)
ap.add_argument("--scene", default="city")
a = ap.parse_args()
if a.scene == "city":
    scene = host.Scene(arrays=S.city_block(a.tris))
else:
    tris, mats = S.cornell_blob(a.tris, 20_000)
    scene = host.Scene(arrays=dict(triangles=tris, materials=mats))
scene.build_bvh()
nodes = scene.arrays()["nodes"].copy()
d = np.asarray((-0.6, -1.5, 3.5), np.float64); d /= np.linalg.norm(d)


def cost(t):
    inner = (t["num_primitives_axis"] >> 16) == 0
    e = np.stack([t["bounds_max"][c].astype(np.float64) - t["bounds_min"][c].astype(np.float64) for c in "xyz"], 1)[inner]
    iso = 0.5 * 0.5 * (e[:, 0] * e[:, 1] + e[:, 1] * e[:, 2] + e[:, 2] * e[:, 0])
    proj = abs(d[0]) * e[:, 1] * e[:, 2] + abs(d[1]) * e[:, 2] * e[:, 0] + abs(d[2]) * e[:, 0] * e[:, 1]
    return float((iso + proj).sum())


t0 = time.time(); ref = own_tree(nodes, 0.5, [d]); t_host = time.time() - t0
c_host, c_ref = cost(ref), cost(nodes)
print("%s, %d nodes: reference topology %.4g, host-built own tree %.4g (%.2f s) = %.3f of the reference's" % (a.scene, len(nodes), c_ref, c_host, t_host, c_host / c_ref), flush=True)
ctx = capi.Context(0)
for frame, stretch in (("world", 1.0), ("light", 1.0), ("light", 2.0), ("light", 4.0), ("light", 8.0)):
    for radius in (8, 16, 32, 64, 128):
        t, sec, rounds = capi.device_tree(ctx, nodes, 0.5, [d], radius=radius, frame_dir=(d if frame == "light" else None), stretch=stretch)
        print("frame %-5s stretch %3.0f radius %3d: device / host %.3f (%.3f of the reference's), %3d rounds, %.3f s" % (frame, stretch, radius, cost(t) / c_host, cost(t) / c_ref, rounds, sec), flush=True)
ctx.close()
