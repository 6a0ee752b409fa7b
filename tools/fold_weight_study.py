"""Which visit probability should the 4-wide fold be optimal for?  (Round 4, analysis; no GPU.)  build_wide_bvh's dynamic programme minimises the sum,
over the records of the fold, of P(record is visited), with P = the root's surface area.  ANY frontier gives bit-identical results (the fold decides
which boxes are tested, never the order of the leaves), so the weights are free.  This tool folds the reference's BVH2 of the benchmark scene with
(a) the surface area, (b) measured box-pass counts of the frame's own closest-hit rays (half of them; the fold is then walked by the OTHER half),
(c) pass counts of a camera-independent proxy population (area-weighted surface origins, cosine directions), and walks each fold with the real
queues (oracle.c: orc_nwide_stats_weighted, exact boxes, nearest slot first).
Result on the 2.8 M-triangle city block, 480x270, 8 bounces: (a) 18.04 visits per ray, (b) 16.76 (-7.1 %, out of sample), (c) 18.2 - 18.5 (worse than
the area): the gain is there, but only for the view's own ray distribution -- a profile-guided re-fold after the first batch, not an upload-time choice.
usage: NT=2800000 python tools/fold_weight_study.py"""
import sys, os, ctypes as C
sys.path.insert(0, '/root/repo'); os.chdir('/root/repo')
import numpy as np
from raytracing_amd import host, scenes as S, types as T
from tests import _oracle
scene = host.Scene(arrays=S.city_block(int(os.environ.get("NT", "2800000")))); scene.add_directional_light((-0.6,-1.5,3.5),(15,10,5))
scene.set_env_path("assets/ibl/CGSkies_0036_free.hdr"); scene.build_bvh(); scene.finalize()
arrays = scene.arrays()
w, h, B = 480, 270, 8
orc = _oracle.Oracle(w, h, arrays); orc.set_camera(host.default_camera(w, h)); orc.set_max_bounces(B)
lib = orc.lib
lib.orc_node_pass_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
lib.orc_nwide_stats_weighted.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
n = w * h
queues = []
orc.stage("reset"); orc.stage("generate_rays")
for bounce in range(B + 1):
    k = int(orc.buffer("ray_counter%d" % (bounce & 1), np.uint32, 1)[0])
    queues.append(orc.buffer("rays%d" % (bounce & 1), T.ray, n)[:k].copy())
    orc.stage("intersect", bounce)
    for st, args in (("shade_miss", (bounce,)), ("clear_counters", (bounce,)), ("shade_hits", (bounce,))):
        orc.stage(st, *args)
    orc.stage("intersect_shadow"); orc.stage("accumulate")
nn = len(arrays["nodes"])
# in-sample: the weights are the pass counts of the very rays the fold is then walked with (an upper bound of what a measured fold can give)
counts = np.zeros(nn, np.float64)
for q in queues:
    q = np.ascontiguousarray(q); lib.orc_node_pass_counts(orc.handle, q.ctypes.data, len(q), counts.ctypes.data)
# out of sample: weights from every second ray, walked with the others
half = np.zeros(nn, np.float64)
for q in queues:
    q2 = np.ascontiguousarray(q[::2]); lib.orc_node_pass_counts(orc.handle, q2.ctypes.data, len(q2), half.ctypes.data)
nodes = arrays["nodes"]
area = np.zeros(nn)
d = [nodes["bounds_max"][c].astype(np.float64) - nodes["bounds_min"][c] for c in "xyz"]
area = d[0]*d[1] + d[1]*d[2] + d[2]*d[0]
def walk(weights, qs):
    tot = np.zeros(6, np.uint64)
    for q in qs:
        q = np.ascontiguousarray(q); c = np.zeros(6, np.uint64)
        if weights is None: lib.orc_nwide_stats.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]; lib.orc_nwide_stats(orc.handle, 4, None, 0, q.ctypes.data, len(q), 0, c.ctypes.data)
        else: lib.orc_nwide_stats_weighted(orc.handle, 4, weights.ctypes.data, q.ctypes.data, len(q), 0, c.ctypes.data)
        rec = c[5]; tot += c; tot[5] = rec
    return tot
for name, wts, qs in (("surface area (production)", None, queues), ("pass counts, in sample", counts + 1e-9 * area / area[0], queues),
                      ("surface area, odd rays", None, [q[1::2] for q in queues]), ("pass counts of the even rays, walked by the odd ones", half + 0.05 * half.sum() / nn * area / area.mean(), [q[1::2] for q in queues])):
    t = walk(wts, qs)
    print("%-55s %8.3f visits per ray, %8.3f steps, %d records" % (name, t[1] / t[0], (t[1] + t[3]) / t[0], t[5]))

# (a) camera-independent proxy population: origins by area over all surfaces, cosine-distributed directions about the normal (both sides)
def proxy(nrays, seed=4):
    rng = np.random.default_rng(seed)
    tris = arrays["triangles"]
    P = np.stack([np.stack([tris[k]["position"][c] for c in "xyz"], 1) for k in ("v1", "v2", "v3")], 1).astype(np.float64)
    e1, e2 = P[:, 1] - P[:, 0], P[:, 2] - P[:, 0]
    nrm = np.cross(e1, e2); ar = 0.5 * np.linalg.norm(nrm, axis=1)
    pick = rng.choice(len(tris), nrays, p=ar / ar.sum())
    u, v = rng.random(nrays), rng.random(nrays)
    flip = u + v > 1; u[flip] = 1 - u[flip]; v[flip] = 1 - v[flip]
    p = P[pick, 0] + u[:, None] * e1[pick] + v[:, None] * e2[pick]
    nn_ = nrm[pick] / np.maximum(np.linalg.norm(nrm[pick], axis=1, keepdims=True), 1e-30)
    nn_ *= np.where(rng.random(nrays) < 0.5, 1.0, -1.0)[:, None]
    r1, r2 = rng.random(nrays), rng.random(nrays)
    r, phi, z = np.sqrt(r1), 2 * np.pi * r2, np.sqrt(1 - r1)
    t1 = np.where(np.abs(nn_[:, :1]) < 0.9, np.array([[1.0, 0, 0]]), np.array([[0, 1.0, 0]]))
    b1 = np.cross(t1, nn_); b1 /= np.linalg.norm(b1, axis=1, keepdims=True); b2 = np.cross(nn_, b1)
    d = r[:, None] * np.cos(phi)[:, None] * b1 + r[:, None] * np.sin(phi)[:, None] * b2 + z[:, None] * nn_
    rays = np.zeros(nrays, T.ray)
    o = p + 1e-3 * nn_
    for i, c in enumerate("xyz"):
        rays["origin"][c] = o[:, i]; rays["direction"][c] = d[:, i]
    rays["origin"]["w"] = 0.0; rays["direction"]["w"] = 20000.0
    return rays
for nrays in (65536, 524288):
    pr = proxy(nrays)
    pc = np.zeros(nn, np.float64)
    lib.orc_node_pass_counts(orc.handle, np.ascontiguousarray(pr).ctypes.data, len(pr), pc.ctypes.data)
    for lam in (0.05, 0.5):
        t = walk(pc + lam * pc.sum() / nn * area / area.mean(), queues)
        print("proxy population, %7d rays, area prior %.2f: %8.3f visits per REAL ray, %8.3f steps, %d records" % (nrays, lam, t[1] / t[0], (t[1] + t[3]) / t[0], t[5]))
