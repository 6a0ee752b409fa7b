"""What the wide-tree walk of k_trace_w4 does per ray on the benchmark scene, counted on the CPU by its restatement in
oracle/oracle.c (orc_wide_trace; tests/test_wide_traversal_oracle.py shows that restatement returns the reference's hits bit
for bit): wide-node visits, leaf arrivals, triangle tests, pushes, pops culled by their entry distance, slots that pass
their box test per visit, deepest stack -- for the closest-hit and the shadow rays of every bounce.  No GPU.
usage: python tools/wide_walk_stats.py [--triangles 2800000] [--width 480 --height 270] [--bounces 8]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from raytracing_amd import host, scenes as S, types as T
from tests import _oracle
from tests.test_wide_bvh import wide_of

ap = argparse.ArgumentParser()
ap.add_argument("--triangles", type=int, default=2_800_000)
ap.add_argument("--width", type=int, default=480)
ap.add_argument("--height", type=int, default=270)
ap.add_argument("--bounces", type=int, default=8)
ap.add_argument("--collapse", type=int, default=1, help="1 = SAH-optimal frontier per record, 2 = two BVH2 levels per record")
a = ap.parse_args()
scene = host.Scene(arrays=S.city_block(a.triangles))
scene.add_directional_light((-0.6, -1.5, 3.5), (15.0, 10.0, 5.0))
scene.set_env_path(os.path.join(ROOT, "assets", "ibl", "CGSkies_0036_free.hdr"))
scene.build_bvh(); scene.finalize()
arrays = scene.arrays()
t0 = time.time()
wide, entry = wide_of(arrays["nodes"], a.collapse)
print("%d triangles, %d BVH2 nodes, %d wide nodes (%.1f s)" % (len(arrays["triangles"]), len(arrays["nodes"]), len(wide), time.time() - t0))
w, h, n = a.width, a.height, a.width * a.height
orc = _oracle.Oracle(w, h, arrays)
orc.set_camera(host.default_camera(w, h)); orc.set_max_bounces(a.bounces)
orc.stage("reset"); orc.stage("generate_rays")
names = _oracle.Oracle.WIDE_COUNTERS
tot = {False: np.zeros(10, np.uint64), True: np.zeros(10, np.uint64)}
fmt = "%-22s %9d rays | per ray: %5.2f wide visits (%.2f of 4 slots pass), %5.2f leaves (%4.1f %% fail their exact box), %5.2f triangles, %5.2f pushes, %5.2f culled pops | deepest stack %d"
def show(tag, c):
    r = max(int(c[0]), 1)
    print(fmt % (tag, c[0], c[1] / r, c[9] / max(int(c[1]), 1), c[2] / r, 100.0 * c[3] / max(int(c[2]), 1), c[4] / r, c[5] / r, c[6] / r, c[7]))
for bounce in range(a.bounces + 1):
    k = int(orc.buffer("ray_counter%d" % (bounce & 1), np.uint32, 1)[0])
    rays = orc.buffer("rays%d" % (bounce & 1), T.ray, n)[:k]
    orc.stage("intersect", bounce)
    c = np.zeros(10, np.uint64)
    got = orc.wide_trace(wide, entry, rays, False, c)
    assert np.array_equal(got["primitive_id"], orc.buffer("hits", T.hit, n)[:k]["primitive_id"])
    show("closest, bounce %d" % bounce, c)
    m = tot[False][7]; tot[False] += c; tot[False][7] = max(m, c[7])
    for st, args in (("shade_miss", (bounce,)), ("clear_counters", (bounce,)), ("shade_hits", (bounce,))):
        orc.stage(st, *args)
    ks = int(orc.buffer("shadow_ray_counter", np.uint32, 1)[0])
    srays = orc.buffer("shadow_rays", T.ray, n)[:ks]
    orc.stage("intersect_shadow")
    c = np.zeros(10, np.uint64)
    assert np.array_equal(orc.wide_trace(wide, entry, srays, True, c), orc.buffer("shadow_hits", np.uint32, n)[:ks])
    show("shadow,  bounce %d" % bounce, c)
    m = tot[True][7]; tot[True] += c; tot[True][7] = max(m, c[7])
    orc.stage("accumulate")
show("closest, all bounces", tot[False]); show("shadow,  all bounces", tot[True])
st = orc.stats()
print("the reference's BVH2 loop on the same rays: %.1f box tests + %.2f triangle tests per closest-hit ray, %.1f + %.2f per shadow ray" % (
    st["closest_nodes"] / max(int(tot[False][0]), 1), st["closest_tris"] / max(int(tot[False][0]), 1),
    st["shadow_nodes"] / max(int(tot[True][0]), 1), st["shadow_tris"] / max(int(tot[True][0]), 1)))
