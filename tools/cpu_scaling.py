"""Thread-scaling probe of the reference-kernel CPU build (oracle/_ref) on this host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import _ref
from raytracing_amd import host, scenes as S, types as T
tris, mats = S.cornell_blob(871200, 20000)
sc = host.Scene(arrays=dict(triangles=tris, materials=mats)); sc.add_directional_light((-0.6, -1.5, 3.5), (15., 10., 5.))
nodes = sc.build_bvh(); sc.set_env_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets/ibl/CGSkies_0036_free.hdr")); sc.finalize()
arr = sc.arrays()
w, h = 640, 360
print("cores", os.cpu_count(), flush=True)
for th in [int(x) for x in sys.argv[1].split(",")]:
    ri = _ref.RefIntegrator(w, h, arr, threads=th); ri.set_camera(T.default_camera(w, h)); ri.set_max_bounces(8)
    ri.integrate(1); r0 = sum(ri.ray_totals()); t = time.time(); ri.integrate(2); dt = time.time() - t
    print("threads %3d: %.2f Mrays/s" % (th, (sum(ri.ray_totals()) - r0) / dt / 1e6), flush=True)
    del ri
