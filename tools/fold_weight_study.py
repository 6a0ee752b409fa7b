"""Which visit probability should the 4-wide fold be optimal for?  (Round 4, analysis on the CPU; no GPU.)  build_wide_bvh's dynamic programme
minimises the sum, over the records of the fold, of P(record is visited), with P = the root's surface area.  ANY frontier gives bit-identical
results (the fold decides which boxes are tested, never the order of the leaves), so the weights are free -- this is the study behind
RT_CTX_OPT_ADAPTIVE_FOLD (rt_hip.hip: FoldAdapt).

default   folds the reference's BVH2 of the benchmark scene with (a) the surface area, (b) measured box-pass counts of the frame's own rays --
          with the ray's initial t_max, and clipped at its hit -- from every 2nd / 8th / 32nd ray and with three strengths of a surface-area prior;
          every fold is then walked by the rays it has NOT seen (oracle.c: orc_nwide_stats_weighted, exact boxes).  Closest-hit and shadow rays.
          Result (2.8 M-triangle city block, 480x270, 8 bounces): 18.04 visits per closest-hit ray on the surface-area fold, 16.76 with pass counts
          (-7.1 %), 16.54 with hit-clipped counts (-8.3 %); 13 700 rays do as well as 220 000 (16.60) and the prior hardly matters; shadow rays
          13.35 -> 11.95 (-10.5 %).  A camera-INDEPENDENT proxy population (surface origins, cosine directions) is worse than the area: 18.2 - 18.5.
--kernel  the production path end to end: rt_debug_adapt_fold (= FoldAdapt's worker) on the rays of a 240x135 probe frame, then k_trace_w4's walk
          restated on the CPU (orc_wide_trace: quantised boxes, shrinking t_max) over a 400x225 frame of the same camera -- hits and verdicts
          equal the reference loop's bit for bit, 18.78 -> 17.35 wide visits per closest-hit ray (steps 21.69 -> 20.26), shadow 14.54 -> 13.24.
          (Measured on the GPU then: closest trace 0.983 -> 0.919 ms per sample, shadow 0.335 -> 0.314, profiles/r04_call27_*.)
--views   four cameras (default; far end looking back; from above; a side street): each view's fold walked by every view's rays.  Own view: -2.4 ...
          -14.2 %; another view's fold: 0 ... +2 % as a rule, +5 % and +11 % in the worst pairs -- hence FoldAdapt adapts again when the camera
          has left the view (fold_view_left).
--tree    beyond the fold (RT_CTX_OPT_ADAPTIVE_FOLD bit 3): for shadow rays the BINARY tree is free too (any tree over the reference's leaves gives an
          any-hit query the reference's verdict), so it is rotated for the probe rays' measured box crossings (csrc/tree_rotate.h) and then folded.
          Steps per unseen shadow ray (400x225 frame; probe 240x135), verdicts asserted equal to the reference loop's for every tree:
          2.8 M triangles: reference topology 14.84, the backend's own tree 13.16 (production), own + rotations 11.43 (-13 %; child<->grandchild
          moves alone 11.98); 300 K triangles: 8.81 / 6.51 / 4.60 (-29 %).  A move has to save 3 % of the crossings at its node: taking every
          small gain locks the search in (5.01 instead of 4.60; a probe twice as large, PROBE_W / PROBE_H, changes nothing).  Dead ends on the way (a per-leaf mass term in own_bvh.h's greedy cost): DESIGN.md section 8.
--order   and the ORDER of a shadow record's slots (bit 4): k_trace_w4<shadow> looks at them as stored and an occluded ray stops at its first hit, so the
          worker stores them likeliest occluder first (by the probe rays' nearest occluders).  The worker's own code path (rt_debug_adapt_shadow_side)
          on the backend's own tree, modes fold / + rotations / + occluder-first: 2.8 M triangles, 25 % of the shadow rays occluded: 13.16 / 11.43 /
          10.05 steps per unseen shadow ray (occluded ones: 21.7 -> 16.2 by the order alone); 300 K triangles, 14 % occluded: 6.51 / 4.60 / 4.30.
usage: NT=2800000 python tools/fold_weight_study.py [--kernel | --views | --tree | --order]     (logs: profiles/r04_fold_weight_study*.log)"""
import sys, os, ctypes as C, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); os.chdir(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracing_amd import host, scenes as S, types as T, capi
from tests import _oracle
from tests.test_wide_bvh import WIDE, wide_of
from tests.test_adaptive_fold import as_probe, same_hits

ap = argparse.ArgumentParser()
ap.add_argument("--kernel", action="store_true"); ap.add_argument("--views", action="store_true"); ap.add_argument("--tree", action="store_true"); ap.add_argument("--order", action="store_true")
args = ap.parse_args()
scene = host.Scene(arrays=S.city_block(int(os.environ.get("NT", "2800000")))); scene.add_directional_light((-0.6, -1.5, 3.5), (15, 10, 5))
scene.set_env_path("assets/ibl/CGSkies_0036_free.hdr"); scene.build_bvh(); scene.finalize()
arrays = scene.arrays()
nodes = arrays["nodes"]; nn = len(nodes)
B = 8


def queues(cam, w, h):
    """per bounce: closest-hit queue, its hits, shadow queue, its verdicts (the oracle, stage by stage)"""
    orc = _oracle.Oracle(w, h, arrays); orc.set_camera(cam); orc.set_max_bounces(B)
    n = w * h; out = []
    orc.stage("reset"); orc.stage("generate_rays")
    for bounce in range(B + 1):
        k = int(orc.buffer("ray_counter%d" % (bounce & 1), np.uint32, 1)[0])
        rays = orc.buffer("rays%d" % (bounce & 1), T.ray, n)[:k].copy()
        orc.stage("intersect", bounce)
        hits = orc.buffer("hits", T.hit, n)[:k].copy()
        for st, a in (("shade_miss", (bounce,)), ("clear_counters", (bounce,)), ("shade_hits", (bounce,))): orc.stage(st, *a)
        ks = int(orc.buffer("shadow_ray_counter", np.uint32, 1)[0])
        srays = orc.buffer("shadow_rays", T.ray, n)[:ks].copy()
        orc.stage("intersect_shadow")
        verdicts = orc.buffer("shadow_hits", np.uint32, n)[:ks].copy()
        orc.stage("accumulate")
        out.append((rays, hits, srays, verdicts))
    return orc, out


def probe_arrays(q, sel=slice(None)):
    o = np.concatenate([as_probe(r[sel], hh[sel])[0] for r, hh, _, _ in q]); d = np.concatenate([as_probe(r[sel])[1] for r, _, _, _ in q])
    so = np.concatenate([as_probe(s[sel])[0] for _, _, s, _ in q]); sd = np.concatenate([as_probe(s[sel])[1] for _, _, s, _ in q])
    return o, d, so, sd


if args.kernel:
    _, qp = queues(T.default_camera(240, 135), 240, 135)                  # what fold_probe traces for a 1080p frame (k = 8)
    o, d, so, sd = probe_arrays(qp)
    print("probe rays: closest-hit %d, shadow %d" % (len(o), len(so)), flush=True)
    t0 = time.time(); rec, entry, roots, cost, adopted = capi.adapt_fold(nodes, o, d)
    print("closest-hit re-fold: %.2f s on this host (the surface-area fold is made first, for its cost), box passes at record roots %.3f -> %.3f, adopted %s" % ((time.time() - t0,) + cost + (adopted,)), flush=True)
    t0 = time.time(); srec, sentry, sroots, scost, sadopted = capi.adapt_fold(nodes, so, sd)
    print("shadow re-fold:      %.2f s, %.3f -> %.3f, adopted %s" % ((time.time() - t0,) + scost + (sadopted,)), flush=True)
    ref_wide, ref_entry = wide_of(nodes, 1)
    cl, sh = rec.view(WIDE).reshape(-1), srec.view(WIDE).reshape(-1)
    orc, q = queues(T.default_camera(400, 225), 400, 225)                 # other pixels, other random numbers
    V = {k: np.zeros(10, np.uint64) for k in ("surface-area fold, closest-hit", "adapted fold, closest-hit", "surface-area fold, shadow", "adapted fold, shadow")}
    for rays, hits, srays, verdicts in q:
        assert same_hits(orc.wide_trace(ref_wide, ref_entry, rays, False, V["surface-area fold, closest-hit"], direct=True), hits) == 0
        assert same_hits(orc.wide_trace(cl, entry, rays, False, V["adapted fold, closest-hit"], direct=True), hits) == 0
        assert np.array_equal(orc.wide_trace(ref_wide, ref_entry, srays, True, V["surface-area fold, shadow"], direct=True), verdicts)
        assert np.array_equal(orc.wide_trace(sh, sentry, srays, True, V["adapted fold, shadow"], direct=True), verdicts)
    for k, v in V.items():
        r = float(v[0])
        print("%-32s rays %9d  wide visits %.3f  leaf arrivals %.3f  triangle tests %.3f  steps %.3f  slots passed %.3f" % (k, v[0], v[1] / r, v[2] / r, v[4] / r, (v[1] + v[4]) / r, v[9] / r), flush=True)
    sys.exit(0)

if args.tree:
    from tests.test_own_tree import own_tree
    ld = np.array([-0.6, -1.5, 3.5]); ld /= np.linalg.norm(ld)
    pw, ph = int(os.environ.get("PROBE_W", "240")), int(os.environ.get("PROBE_H", "135"))      # (a larger probe: does the search fit its sample less?)
    _, qp = queues(T.default_camera(pw, ph), pw, ph)
    _, _, so, sd = probe_arrays(qp)
    print("shadow probe rays", len(so), "of a %dx%d probe frame" % (pw, ph), flush=True)
    trees = {"reference topology": nodes, "the backend's own tree (production)": own_tree(nodes, 0.5, [ld])}
    for base in list(trees):
        for moves, mg in ((1, 0.0), (3, 0.0), (3, 0.03), (3, 0.1)):
            t0 = time.time(); rot, crossings, made = capi.rotate_tree(trees[base], so, sd, 8, moves, mg)
            print("%s, moves %d, min gain %.2f: %d rotations in %.1f s, interior boxes crossed per probe ray %.2f -> %.2f" % (base, moves, mg, made, time.time() - t0, crossings[0], crossings[1]), flush=True)
            trees[base + " + rotations (moves %d, min gain %.2f)" % (moves, mg)] = rot
    orc, q = queues(T.default_camera(400, 225), 400, 225)
    for name, tree in trees.items():
        rec, entry, roots, cost, adopted = capi.adapt_fold(tree, so, sd)
        wide = rec.view(WIDE).reshape(-1)
        V = np.zeros(10, np.uint64)
        for rays, hits, srays, verdicts in q:
            assert np.array_equal(orc.wide_trace(wide, entry, srays, True, V, direct=True), verdicts)
        r = float(V[0])
        print("%-72s adapted fold: box passes at record roots per probe ray %.2f -> %.2f | unseen shadow rays: wide visits %.3f leaf arrivals %.3f triangle tests %.3f steps %.3f" % (
            name, cost[0], cost[1], V[1] / r, V[2] / r, V[4] / r, (V[1] + V[4]) / r), flush=True)
    sys.exit(0)

if args.order:
    from tests.test_own_tree import own_tree
    ld = np.array([-0.6, -1.5, 3.5]); ld /= np.linalg.norm(ld)
    _, qp = queues(T.default_camera(240, 135), 240, 135)
    _, _, so, sd = probe_arrays(qp)
    vp = np.concatenate([v for _, _, _, v in qp])
    print("shadow probe rays %d, %.1f %% of them occluded" % (len(so), 100.0 * (vp != 0xFFFFFFFF).mean()), flush=True)
    own = own_tree(nodes, 0.5, [ld])
    orc, q = queues(T.default_camera(400, 225), 400, 225)
    for name, mode in (("fold adapted (production default)", 5), ("fold adapted, slots occluder first", 21), ("tree rotated + fold adapted", 13), ("tree rotated + fold adapted, slots occluder first", 29)):
        t0 = time.time()
        rec, entry, roots, tree, cost, made, adopted, moved = capi.adapt_shadow_side(own, so, sd, mode, arrays["triangles"])
        dt = time.time() - t0
        wide = rec.view(WIDE).reshape(-1)
        V, Vo = np.zeros(10, np.uint64), np.zeros(10, np.uint64)
        for rays, hits, srays, verdicts in q:
            assert np.array_equal(orc.wide_trace(wide, entry, srays, True, V, direct=True), verdicts), name
            orc.wide_trace(wide, entry, srays[verdicts != 0xFFFFFFFF], True, Vo, direct=True)
        r, ro = float(V[0]), float(Vo[0])
        print("%-52s %5.1f s on this host, %5d rotations, %6d records reordered | unseen shadow rays: steps %.3f (wide visits %.3f); the occluded %.0f %% of them: steps %.3f" % (
            name, dt, made, moved, (V[1] + V[4]) / r, V[1] / r, 100 * ro / r, (Vo[1] + Vo[4]) / ro), flush=True)
    sys.exit(0)

if args.views:
    mn = [float(nodes["bounds_min"][c][0]) for c in "xyz"]; mx = [float(nodes["bounds_max"][c][0]) for c in "xyz"]
    print("scene bounds", mn, mx, flush=True)

    def cam_at(pos, front, w, h):
        cam = T.default_camera(w, h).copy()
        f = np.asarray(front, np.float64); f /= np.linalg.norm(f)
        r = np.cross(f, [0, 0, 1.0]); r /= np.linalg.norm(r); u = np.cross(r, f)
        for i, k in enumerate("xyz"):
            cam["position"][k] = pos[i]; cam["front"][k] = f[i]; cam["up"][k] = u[i]
        return cam
    cx, cy = 0.5 * (mn[0] + mx[0]), 0.5 * (mn[1] + mx[1])
    cams = {"A default": T.default_camera(240, 135),
            "B far end, looking back": cam_at((cx + 0.3 * (mx[0] - mn[0]), mx[1] - 0.05 * (mx[1] - mn[1]), 1.5), (-0.3, -1.0, -0.05), 240, 135),
            "C from above, looking down": cam_at((cx, cy, mx[2] + 0.5 * (mx[1] - mn[1])), (0.05, 0.1, -1.0), 240, 135),
            "D side street": cam_at((mn[0] + 0.2 * (mx[0] - mn[0]), cy, 2.0), (1.0, 0.2, 0.0), 240, 135)}
    Q = {k: queues(c, 240, 135) for k, c in cams.items()}
    folds = {"surface area": wide_of(nodes, 1)}
    for k, (orc, q) in Q.items():
        o, d, _, _ = probe_arrays(q, slice(0, None, 2))
        rec, entry, roots, cost, adopted = capi.adapt_fold(nodes, o, d)
        folds["adapted to " + k[0]] = (rec.view(WIDE).reshape(-1), entry)
        print("%-28s %7d probe rays (the even ones), box passes at record roots %.2f -> %.2f" % ((k, len(o)) + cost), flush=True)
    print("%-30s" % "wide visits per ray (odd rays)" + "".join("%-22s" % f for f in folds))
    for k, (orc, q) in Q.items():
        row = []
        for wide, entry in folds.values():
            V = np.zeros(10, np.uint64)
            for rays, hits, srays, verdicts in q:
                orc.wide_trace(wide, entry, rays[1::2], False, V, direct=True)
            row.append(float(V[1]) / float(V[0]))
        print("%-30s" % k + "".join("%-22s" % ("%.3f (%+.1f %%)" % (v, 100 * (v / row[0] - 1))) for v in row), flush=True)
    sys.exit(0)

# ---- default: which weights, how many rays, how much prior --------------------------------------------------------------------------------
w, h = 480, 270
orc, q = queues(T.default_camera(w, h), w, h)
lib = orc.lib
lib.orc_node_pass_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
lib.orc_nwide_stats_weighted.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
lib.orc_nwide_stats.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
queues_cl = [r for r, _, _, _ in q]; queues_sh = [s for _, _, s, _ in q]
clipped = []
for rays, hits, _, _ in q:
    qc = rays.copy()
    hit = hits["primitive_id"] != 0xFFFFFFFF
    qc["direction"]["w"] = np.where(hit, hits["t"] * np.float32(1.0001), rays["direction"]["w"]).astype(np.float32)
    clipped.append(qc)
d3 = [nodes["bounds_max"][c].astype(np.float64) - nodes["bounds_min"][c] for c in "xyz"]
area = d3[0] * d3[1] + d3[1] * d3[2] + d3[2] * d3[0]


def counts_of(qs, sel):
    c = np.zeros(nn, np.float64)
    for x in qs:
        x2 = np.ascontiguousarray(x[sel]); lib.orc_node_pass_counts(orc.handle, x2.ctypes.data, len(x2), c.ctypes.data)
    return c


def walk(weights, qs, shadow=0):
    tot = np.zeros(6, np.uint64)
    for x in qs:
        x = np.ascontiguousarray(x); c = np.zeros(6, np.uint64)
        if len(x) == 0: continue
        if weights is None: lib.orc_nwide_stats(orc.handle, 4, None, 0, x.ctypes.data, len(x), shadow, c.ctypes.data)
        else: lib.orc_nwide_stats_weighted(orc.handle, 4, weights.ctypes.data, x.ctypes.data, len(x), shadow, c.ctypes.data)
        rec = c[5]; tot += c; tot[5] = rec
    return tot


def show(name, t):
    print("%-78s %8.3f visits per ray, %8.3f steps, %d records" % (name, t[1] / t[0], (t[1] + t[3]) / t[0], t[5]), flush=True)


prior = lambda c, lam: c + lam * c.sum() / nn * area / area.mean()
odd = [x[1::2] for x in queues_cl]
show("closest-hit: surface area (production before the adaptation), odd rays", walk(None, odd))
for step in (2, 8, 32):
    sel = slice(0, None, step)                                            # even indices only: out of sample for the odd rays
    c_init, c_clip = counts_of(queues_cl, sel), counts_of(clipped, sel)
    nr = sum(len(x[sel]) for x in queues_cl)
    for lam in (0.01, 0.05, 0.3):
        show("closest-hit: box passes with the initial t_max, every %dth ray (%d), prior %.2f" % (step, nr, lam), walk(prior(c_init, lam), odd))
        show("closest-hit: box passes clipped at the hit,     every %dth ray (%d), prior %.2f" % (step, nr, lam), walk(prior(c_clip, lam), odd))
sodd = [x[1::2] for x in queues_sh]
show("shadow: surface-area fold of the reference tree, odd rays", walk(None, sodd, 1))
for step in (2, 8):
    cs = counts_of(queues_sh, slice(0, None, step))
    for lam in (0.05, 0.3):
        show("shadow: box passes of every %dth shadow ray, prior %.2f" % (step, lam), walk(prior(cs, lam), sodd, 1))


def proxy(nrays, seed=4):
    """camera-independent: origins by area over all surfaces, cosine-distributed directions about the normal (both sides)"""
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
    dd = r[:, None] * np.cos(phi)[:, None] * b1 + r[:, None] * np.sin(phi)[:, None] * b2 + z[:, None] * nn_
    rays = np.zeros(nrays, T.ray)
    oo = p + 1e-3 * nn_
    for i, c in enumerate("xyz"):
        rays["origin"][c] = oo[:, i]; rays["direction"][c] = dd[:, i]
    rays["origin"]["w"] = 0.0; rays["direction"]["w"] = 20000.0
    return rays


for nrays in (65536, 524288):
    pc = counts_of([proxy(nrays)], slice(None))
    for lam in (0.05, 0.5):
        show("closest-hit: camera-independent proxy population, %7d rays, prior %.2f, walked by the REAL rays" % (nrays, lam), walk(prior(pc, lam), queues_cl))
