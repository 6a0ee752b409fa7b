"""What a tree of the backend's own buys, counted on the CPU before any GPU time is spent (round 4).

The reference's BVH2 (src/bvh.cpp: one split axis, 12 buckets) fixes, for an any-hit query, only its LEAVES; own_bvh.h builds
another binary tree over those leaves (full-sweep SAH on all axes; for shadow rays on the projected area along the directional
lights), build_wide_bvh folds either into the 4-wide records k_trace_w4 walks, and oracle.c's restatement of that walk counts the
steps.  Shadow verdicts must equal the reference's bit for bit (asserted); closest hits on the own tree are the TOLERANCE mode:
the differing hits are counted.
usage: python tools/own_tree_study.py [--triangles 2800000] [--width 480 --height 270] [--bounces 8] [--scene city|foliage|dragon]"""
import argparse, ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from raytracing_amd import capi, host, scenes as S, types as T
from tests import _oracle
from tests.test_wide_bvh import wide_of, WIDE

ap = argparse.ArgumentParser()
ap.add_argument("--triangles", type=int, default=2_800_000)
ap.add_argument("--width", type=int, default=480)
ap.add_argument("--height", type=int, default=270)
ap.add_argument("--bounces", type=int, default=8)
ap.add_argument("--scene", default="city")
ap.add_argument("--iso", type=float, default=0.05, help="isotropic share of the shadow metric")
a = ap.parse_args()
LIGHT = (-0.6, -1.5, 3.5)


def own_tree(nodes, iso, dirs):
    lib = capi.load()
    nodes = np.ascontiguousarray(nodes)
    d = np.ascontiguousarray(np.asarray(dirs, np.float32).reshape(-1, 3))
    n = C.c_uint32()
    if lib.rt_debug_own_bvh(nodes.ctypes.data, len(nodes), iso, d.ctypes.data if len(d) else None, len(d), None, 0, C.byref(n)):
        raise RuntimeError(lib.rt_last_error(None).decode())
    out = np.zeros(n.value, nodes.dtype)
    assert lib.rt_debug_own_bvh(nodes.ctypes.data, len(nodes), iso, d.ctypes.data if len(d) else None, len(d), out.ctypes.data, len(out), C.byref(n)) == 0
    return out


def wide_metric(nodes, iso, dirs):
    lib = capi.load()
    nodes = np.ascontiguousarray(nodes)
    d = np.ascontiguousarray(np.asarray(dirs, np.float32).reshape(-1, 3))
    n, entry = C.c_uint32(), C.c_uint32()
    if lib.rt_debug_wide_bvh_metric(nodes.ctypes.data, len(nodes), iso, d.ctypes.data if len(d) else None, len(d), None, 0, C.byref(n), C.byref(entry)):
        raise RuntimeError(lib.rt_last_error(None).decode())
    out = np.zeros(n.value, WIDE)
    assert lib.rt_debug_wide_bvh_metric(nodes.ctypes.data, len(nodes), iso, d.ctypes.data if len(d) else None, len(d), out.ctypes.data, len(out), C.byref(n), C.byref(entry)) == 0
    return out, entry.value


if __name__ == "__main__":
    if a.scene == "dragon":
        tris, mats = S.cornell_blob(a.triangles, 20_000)
        scene = host.Scene(arrays=dict(triangles=tris, materials=mats))
    else:
        scene = host.Scene(arrays={"city": S.city_block, "foliage": S.dense_foliage}[a.scene](a.triangles))
    scene.add_directional_light(LIGHT, (15.0, 10.0, 5.0))
    scene.set_env_path(os.path.join(ROOT, "assets", "ibl", "CGSkies_0036_free.hdr"))
    scene.build_bvh(); scene.finalize()
    arrays = scene.arrays()
    nodes = arrays["nodes"]
    ld = np.asarray(LIGHT, np.float64); ld /= np.linalg.norm(ld)
    trees = {}
    t0 = time.time(); trees["reference topology"] = wide_of(nodes, 1); t_ref = time.time() - t0
    t0 = time.time(); own_iso = own_tree(nodes, 1.0, []); trees["own, surface area"] = wide_metric(own_iso, 1.0, []); t_iso = time.time() - t0
    t0 = time.time(); own_dir = own_tree(nodes, a.iso, [ld]); trees["own, projected area along the light"] = wide_metric(own_dir, a.iso, [ld]); t_dir = time.time() - t0
    t0 = time.time()
    rec, ent, report = capi.choose_tree(arrays, True, 1)
    trees["what rt_scene_upload picks for shadow rays"] = (rec.view(WIDE).reshape(-1), ent)
    rec, ent, report2 = capi.choose_tree(arrays, False, 1)
    trees["what it picks for closest-hit (tolerance mode)"] = (rec.view(WIDE).reshape(-1), ent)
    print(report + report2 + "(both choices: %.1f s)" % (time.time() - t0))
    print("%d triangles, %d BVH2 nodes (%d leaves); wide records: %s; build seconds: collapse of the reference tree %.1f, own trees %.1f / %.1f" % (
        len(arrays["triangles"]), len(nodes), int(((nodes["num_primitives_axis"] >> 16) != 0).sum()),
        ", ".join("%s %d" % (k, len(v[0])) for k, v in trees.items()), t_ref, t_iso, t_dir))
    w, h, n = a.width, a.height, a.width * a.height
    orc = _oracle.Oracle(w, h, arrays)
    orc.set_camera(host.default_camera(w, h)); orc.set_max_bounces(a.bounces)
    orc.stage("reset"); orc.stage("generate_rays")
    tot = {(k, s): np.zeros(10, np.uint64) for k in trees for s in (False, True)}
    widths = (4, 8, 16)
    nw = {(t, wd, sh): dict(rays=0, visits=0, leaf_arrivals=0, triangle_tests=0, slots_tested=0, records=0) for t in ("reference topology", "own, surface area") for wd in widths for sh in (False, True)}
    nw_nodes = {"reference topology": None, "own, surface area": own_iso}
    dist_tot = {k: np.zeros(10, np.uint64) for k in trees}                  # closest-hit rays, slots visited by entry distance
    dist_differ = {k: 0 for k in trees}
    differ = {k: 0 for k in trees}
    differ_t = {k: 0 for k in trees}
    for bounce in range(a.bounces + 1):
        k = int(orc.buffer("ray_counter%d" % (bounce & 1), np.uint32, 1)[0])
        rays = orc.buffer("rays%d" % (bounce & 1), T.ray, n)[:k].copy()
        orc.stage("intersect", bounce)
        want = orc.buffer("hits", T.hit, n)[:k].copy()
        for name, (wide, entry) in trees.items():
            c = np.zeros(10, np.uint64)
            got = orc.wide_trace(wide, entry, rays, False, c, direct=True)
            m = tot[(name, False)][7]; tot[(name, False)] += c; tot[(name, False)][7] = max(m, c[7])
            hit = want["primitive_id"] != 0xFFFFFFFF
            bad = got["primitive_id"] != want["primitive_id"]
            bad |= hit & ((got["t"] != want["t"]) | (np.ascontiguousarray(got["bc"]).view(np.float32).reshape(-1, 2) != np.ascontiguousarray(want["bc"]).view(np.float32).reshape(-1, 2)).any(1))
            differ[name] += int(bad.sum())
            differ_t[name] += int((hit & (got["t"] != want["t"])).sum())
            c = np.zeros(10, np.uint64)
            gd = orc.wide_trace(wide, entry, rays, False, c, direct=True, by_distance=True)
            dist_tot[name] += c
            dist_differ[name] += int(((gd["primitive_id"] != want["primitive_id"]) | (hit & (gd["t"] != want["t"]))).sum())
        for tname, nd in nw_nodes.items():
            for wd in widths:
                r_ = orc.nwide_stats(wd, rays, False, nd)
                for key, val in r_.items():
                    nw[(tname, wd, False)][key] = val if key == "records" else nw[(tname, wd, False)][key] + val
        for st, args in (("shade_miss", (bounce,)), ("clear_counters", (bounce,)), ("shade_hits", (bounce,))):
            orc.stage(st, *args)
        ks = int(orc.buffer("shadow_ray_counter", np.uint32, 1)[0])
        srays = orc.buffer("shadow_rays", T.ray, n)[:ks].copy()
        orc.stage("intersect_shadow")
        swant = orc.buffer("shadow_hits", np.uint32, n)[:ks].copy()
        for name, (wide, entry) in trees.items():
            c = np.zeros(10, np.uint64)
            got = orc.wide_trace(wide, entry, srays, True, c, direct=True)
            assert np.array_equal(got, swant), "shadow verdicts differ on the tree '%s' at bounce %d" % (name, bounce)
            m = tot[(name, True)][7]; tot[(name, True)] += c; tot[(name, True)][7] = max(m, c[7])
        for tname, nd in nw_nodes.items():
            for wd in widths:
                r_ = orc.nwide_stats(wd, srays, True, nd)
                for key, val in r_.items():
                    nw[(tname, wd, True)][key] = val if key == "records" else nw[(tname, wd, True)][key] + val
        orc.stage("accumulate")
    print("%-48s %-8s %10s %8s %8s %8s %8s %8s %6s | closest hits that differ from the reference's" % ("tree", "rays", "count", "visits", "leaves", "tris", "steps", "pushes", "stack"))
    for (name, sh), c in tot.items():
        r = max(int(c[0]), 1)
        steps = (int(c[1]) + int(c[4]) + int(c[3])) / r          # wide-node visits + leaf passes (a failed box test is a pass too)
        print("%-48s %-8s %10d %8.2f %8.2f %8.2f %8.2f %8.2f %6d%s" % (name, "shadow" if sh else "closest", c[0], c[1] / r, c[2] / r, c[4] / r, steps, c[5] / r, c[7],
            "" if sh else " | %d of %d (%.2e), of which %d in t" % (differ[name], c[0], differ[name] / r, differ_t[name])))
    print("closest-hit rays with the slots of a record visited by ENTRY DISTANCE instead of the order table (5 compare-exchanges on computed keys "
          "instead of 4 tabulated ones: ~ +15 vector instructions per visit):")
    for name, c in dist_tot.items():
        r = max(int(c[0]), 1)
        print("%-48s %8.2f visits %8.2f steps | %d hits differ" % (name, c[1] / r, (int(c[1]) + int(c[4]) + int(c[3])) / r, dist_differ[name]))
    print("wider records, priced on the same queues (SAH-optimal fold for each width, EXACT boxes, closest-hit slots nearest first; 4-wide here = the "
          "production fold without its 8-bit rounding and with distance order): record visits + triangle passes per ray, slots tested per ray")
    for (tname, wd, sh), c in nw.items():
        r = max(c["rays"], 1)
        print("%-22s %2d-wide %-8s %8.2f visits %8.2f steps %8.1f slot tests per ray   %9d records" % (
            tname, wd, "shadow" if sh else "closest", c["visits"] / r, (c["visits"] + c["triangle_tests"]) / r, c["slots_tested"] / r, c["records"]))
