"""The trees of the backend's own (raytracing_amd/csrc/own_bvh.h, tree_select.h; round 4), checked ON THE CPU.

The reference's any-hit query (trace_bvh.cl -DSHADOW_RAYS, :107-109,164-167) never shrinks t_max, so its verdict is an OR over
the LEAVES whose exact box passes -- whatever sits above them.  own_bvh.h builds another binary tree over exactly those leaves;
build_wide_bvh folds it into the records k_trace_w4 walks.  Shown here:
  * the own tree is a tree in the reference's linear layout over exactly the reference's leaves, interior boxes exact unions,
    and its 4-wide fold has every invariant the reference's fold has (tests/test_wide_bvh.check);
  * the kernel's walk (restated in oracle/oracle.c) over the own tree returns the reference loop's SHADOW verdict for every
    shadow ray a path tracer produces, bit for bit, on golden scenes, a city block, a dense mesh and random soups;
  * closest-hit rays on an own tree (the opt-in TOLERANCE mode) find the reference's hit except where candidates tie;
  * rt_scene_upload's choice (rt_debug_choose_tree) is one of the two trees, and says what it measured."""
import ctypes as C
import os
import numpy as np
import pytest
from tests import _oracle
from tests.test_wide_bvh import WIDE, check, wide_of
from raytracing_amd import capi, host, scenes as S, types as T

LIGHT = ((-0.6, -1.5, 3.5), (15.0, 10.0, 5.0))


def own_tree(nodes, iso=1.0, dirs=()):
    lib = capi.load()
    nodes = np.ascontiguousarray(nodes)
    d = np.ascontiguousarray(np.asarray(dirs, np.float32).reshape(-1, 3))
    n = C.c_uint32()
    if lib.rt_debug_own_bvh(nodes.ctypes.data, len(nodes), iso, d.ctypes.data if len(d) else None, len(d), None, 0, C.byref(n)):
        raise capi.RtError(lib.rt_last_error(None).decode())
    out = np.zeros(n.value, nodes.dtype)
    assert lib.rt_debug_own_bvh(nodes.ctypes.data, len(nodes), iso, d.ctypes.data if len(d) else None, len(d), out.ctypes.data, len(out), C.byref(n)) == 0
    return out


def wide_metric(nodes, iso=1.0, dirs=()):
    lib = capi.load()
    nodes = np.ascontiguousarray(nodes)
    d = np.ascontiguousarray(np.asarray(dirs, np.float32).reshape(-1, 3))
    n, entry = C.c_uint32(), C.c_uint32()
    if lib.rt_debug_wide_bvh_metric(nodes.ctypes.data, len(nodes), iso, d.ctypes.data if len(d) else None, len(d), None, 0, C.byref(n), C.byref(entry)):
        raise capi.RtError(lib.rt_last_error(None).decode())
    out = np.zeros(n.value, WIDE)
    assert lib.rt_debug_wide_bvh_metric(nodes.ctypes.data, len(nodes), iso, d.ctypes.data if len(d) else None, len(d), out.ctypes.data, len(out), C.byref(n), C.byref(entry)) == 0
    return out, entry.value


def light_dir():
    d = np.asarray(LIGHT[0], np.float64)
    return d / np.linalg.norm(d)


def _finish(scene, env_map, point=True):
    scene.add_directional_light(*LIGHT)
    if point:
        scene.add_point_light((0.3, 0.8, 1.6), (4.0, 4.0, 3.0))
    scene.build_bvh()
    scene.set_env_image(env_map)
    scene.finalize()
    return scene.arrays()


def check_own_structure(nodes, own):
    """`own` is a binary tree in the reference's linear layout over exactly the leaves of `nodes`, boxes exact unions."""
    is_leaf = lambda a: (a["num_primitives_axis"] >> 16) != 0
    ref_leaves = nodes[is_leaf(nodes)]
    own_leaves = own[is_leaf(own)]
    assert len(own) == 2 * len(ref_leaves) - 1
    key = lambda a: sorted(zip(a["offset"].tolist(), (a["num_primitives_axis"] >> 16).tolist(), [x.tobytes() for x in a["bounds_min"]],
                               [x.tobytes() for x in a["bounds_max"]]))
    assert key(own_leaves) == key(ref_leaves)                             # the same leaves: first triangle, count, exact box
    bmin = np.stack([own["bounds_min"][c] for c in "xyz"], 1)
    bmax = np.stack([own["bounds_max"][c] for c in "xyz"], 1)
    size = np.zeros(len(own), np.int64)                                   # nodes in the subtree
    for i in range(len(own) - 1, -1, -1):
        if is_leaf(own[i:i + 1])[0]:
            size[i] = 1
            continue
        a, b = i + 1, int(own["offset"][i])
        assert i + 1 < b < len(own) and (int(own["num_primitives_axis"][i]) & 0xFFFF) <= 2
        assert b == a + size[a]                                           # the second child follows the first child's subtree
        size[i] = 1 + size[a] + size[b]
        assert np.array_equal(bmin[i], np.minimum(bmin[a], bmin[b])) and np.array_equal(bmax[i], np.maximum(bmax[a], bmax[b]))
    assert size[0] == len(own)


def shadow_and_closest_on(arrays, trees, w, h, bounces, samples=1):
    """Oracle stage by stage; every shadow queue is traced by the reference loop and by the wide walk over each tree of `trees`
    (name -> (records, entry)): verdicts must be equal.  Returns per tree (closest rays, closest hits that differ, counters)."""
    orc = _oracle.Oracle(w, h, arrays)
    orc.set_camera(T.default_camera(w, h))
    orc.set_max_bounces(bounces)
    n = w * h
    out = {k: dict(rays=0, differ=0, shadow=np.zeros(10, np.uint64), closest=np.zeros(10, np.uint64)) for k in trees}
    for _ in range(samples):
        orc.stage("reset") if orc.sample_count() == 0 else None
        orc.stage("generate_rays")
        for bounce in range(bounces + 1):
            k = int(orc.buffer("ray_counter%d" % (bounce & 1), np.uint32, 1)[0])
            rays = orc.buffer("rays%d" % (bounce & 1), T.ray, n)[:k].copy()
            orc.stage("intersect", bounce)
            want = orc.buffer("hits", T.hit, n)[:k].copy()
            hit = want["primitive_id"] != 0xFFFFFFFF
            for name, (wide, entry) in trees.items():
                got = orc.wide_trace(wide, entry, rays, False, out[name]["closest"], direct=True)
                bad = got["primitive_id"] != want["primitive_id"]
                bad |= hit & (got["t"] != want["t"])
                bad |= hit & (np.ascontiguousarray(got["bc"]).view(np.float32).reshape(-1, 2) != np.ascontiguousarray(want["bc"]).view(np.float32).reshape(-1, 2)).any(1)
                out[name]["rays"] += k
                out[name]["differ"] += int(bad.sum())
            for st, args in (("shade_miss", (bounce,)), ("clear_counters", (bounce,)), ("shade_hits", (bounce,))):
                orc.stage(st, *args)
            ks = int(orc.buffer("shadow_ray_counter", np.uint32, 1)[0])
            srays = orc.buffer("shadow_rays", T.ray, n)[:ks].copy()
            orc.stage("intersect_shadow")
            swant = orc.buffer("shadow_hits", np.uint32, n)[:ks].copy()
            for name, (wide, entry) in trees.items():
                for direct in (False, True):
                    got = orc.wide_trace(wide, entry, srays, True, out[name]["shadow"] if direct else None, direct=direct)
                    assert np.array_equal(got, swant), ("shadow verdicts differ", name, bounce, direct)
            orc.stage("accumulate")
        orc.stage("advance")
    return out


def trees_of(arrays):
    nodes = arrays["nodes"]
    d = light_dir()
    own_sa, own_dir = own_tree(nodes), own_tree(nodes, 0.5, [d])
    check_own_structure(nodes, own_sa)
    check_own_structure(nodes, own_dir)
    return {"reference": wide_of(nodes, 1), "own, surface area": wide_metric(own_sa), "own, projected area": wide_metric(own_dir, 0.5, [d])}, (own_sa, own_dir)


def test_own_tree_structure_and_its_wide_fold_on_the_golden_scenes(golden_scenes):
    for name in ("cornell", "coverage"):
        nodes = golden_scenes[name]["nodes"]
        for own in (own_tree(nodes), own_tree(nodes, 0.05, [light_dir()]), own_tree(nodes, 0.0, [(0.0, 0.0, 1.0)])):
            check_own_structure(nodes, own)
            check(own, 1)                                                 # every invariant of the 4-wide fold (tests/test_wide_bvh.py)
            check(own, 2)


def test_shadow_verdicts_on_own_trees_equal_the_reference_loop_golden_scenes(golden_scenes):
    for name, (w, h, b) in {"cornell": (64, 48, 5), "coverage": (72, 56, 7)}.items():
        trees, _ = trees_of(golden_scenes[name])
        out = shadow_and_closest_on(golden_scenes[name], trees, w, h, b, samples=2)
        assert out["reference"]["differ"] == 0 and out["reference"]["shadow"][0] > 0


def test_shadow_verdicts_on_a_city_block_and_what_the_own_tree_saves(env_map):
    arrays = _finish(host.Scene(arrays=S.city_block(60_000)), env_map)
    trees, (own_sa, _) = trees_of(arrays)
    check(own_sa, 1)
    out = shadow_and_closest_on(arrays, trees, 96, 54, 8)
    steps = lambda c: (int(c[1]) + int(c[4]) + int(c[3])) / max(int(c[0]), 1)
    ref, sa = steps(out["reference"]["shadow"]), steps(out["own, surface area"]["shadow"])
    assert sa < 1.05 * ref                                                # not worse by more than noise; rt_scene_upload measures before it switches
    # closest-hit rays on the own trees: the TOLERANCE mode -- the reference's hit except where candidates tie
    for name in ("own, surface area", "own, projected area"):
        assert out[name]["differ"] <= 1e-3 * out[name]["rays"], (name, out[name]["differ"], out[name]["rays"])


def test_shadow_verdicts_on_a_dense_mesh(env_map):
    tris, mats = S.cornell_blob(30_000, 3_000)
    arrays = _finish(host.Scene(arrays=dict(triangles=tris, materials=mats)), env_map)
    trees, _ = trees_of(arrays)
    out = shadow_and_closest_on(arrays, trees, 80, 60, 6)
    steps = lambda c: (int(c[1]) + int(c[4]) + int(c[3])) / max(int(c[0]), 1)
    # the Cornell shell + blob is where the projected-area metric pays (-33 % at 0.9 M triangles, tools/own_tree_study.py)
    assert steps(out["own, projected area"]["shadow"]) < steps(out["reference"]["shadow"])


@pytest.mark.parametrize("seed", range(6))
def test_shadow_verdicts_on_random_soups(seed, env_map):
    """Slivers, coincident triangles (ties!), lights inside the geometry, an axis-parallel light (those rays take the BVH2 walk)."""
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(2, 600))
    P = rng.normal(size=(n, 1, 3)) * 1.2 + rng.normal(size=(n, 3, 3)) * float(10.0 ** rng.uniform(-1.2, 0.0)) + np.array([0.0, 2.5, 1.0])
    P = P.astype(np.float32)
    if seed % 2:
        P[: n // 4] = P[0]
    N = np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 0])
    N = (N / np.maximum(np.linalg.norm(N, axis=1, keepdims=True), 1e-20)).astype(np.float32)[:, None, :].repeat(3, 1)
    tris = S.to_triangles([(P, N, np.zeros((n, 3, 2), np.float32), 0)])
    mats = np.array([S.make_material(kd=(0.7, 0.6, 0.5), ks=(0.3, 0.3, 0.3), roughness=0.3)], dtype=T.packed_material)
    s = host.Scene(arrays=dict(triangles=tris, materials=mats))
    if seed == 3:
        s.add_directional_light((0.0, 0.0, 1.0), (5.0, 5.0, 5.0))
    arrays = _finish(s, env_map)
    if (arrays["nodes"]["num_primitives_axis"][0] >> 16) != 0:
        pytest.skip("a single leaf: no tree to build")
    trees, _ = trees_of(arrays)
    out = shadow_and_closest_on(arrays, trees, 48, 40, 5)
    assert out["reference"]["differ"] == 0 and out["reference"]["shadow"][0] > 0


def test_own_tree_on_a_lopsided_scene(env_map):
    """Triangles whose sizes and spacings grow geometrically along a line: every SAH split peels off one end, the kind of input that
    makes a top-down builder recurse as deep as it has leaves.  own_bvh.h recurses into the smaller part only; the tree it builds
    is valid, and too deep a fold falls back to the reference topology instead of failing the upload."""
    n = 6000
    k = np.arange(n, dtype=np.float64)
    x = 1.0005 ** k * 1e-3 * (1.0 + k)                                      # monotone, geometric spacing
    P = np.zeros((n, 3, 3), np.float64)
    P[:, 0] = np.stack([x, np.zeros(n), np.zeros(n)], 1)
    P[:, 1] = P[:, 0] + np.stack([1e-4 * (1.0 + k), np.zeros(n), np.zeros(n)], 1)
    P[:, 2] = P[:, 0] + np.stack([np.zeros(n), 1e-4 * (1.0 + k), np.full(n, 1e-4)], 1)
    P = P.astype(np.float32)
    N = np.zeros((n, 3, 3), np.float32); N[..., 2] = 1.0
    tris = S.to_triangles([(P, N, np.zeros((n, 3, 2), np.float32), 0)])
    mats = np.array([S.make_material(kd=(0.7, 0.6, 0.5))], dtype=T.packed_material)
    arrays = _finish(host.Scene(arrays=dict(triangles=tris, materials=mats)), env_map, point=False)
    own = own_tree(arrays["nodes"])
    check_own_structure(arrays["nodes"], own)
    rec, entry, report = capi.choose_tree(arrays, True, 1)                  # never raises: an own tree that does not qualify is just not taken
    assert "shadow tree" in report


def test_rt_scene_upload_measures_before_it_switches(env_map):
    tris, mats = S.cornell_blob(30_000, 3_000)
    arrays = _finish(host.Scene(arrays=dict(triangles=tris, materials=mats)), env_map, point=False)
    ref, ref_entry = wide_of(arrays["nodes"], 1)
    for shadow in (True, False):
        rec, entry, report = capi.choose_tree(arrays, shadow, 1)
        assert ("shadow tree" if shadow else "closest-hit tree") in report and "steps per proxy ray" in report
        picked_own = "-> own" in report
        assert picked_own != (rec.tobytes() == ref.tobytes())             # either the reference's records, or other ones
        forced, _, rep2 = capi.choose_tree(arrays, shadow, 2)
        assert "forced" in rep2 and forced.tobytes() != ref.tobytes()
        if picked_own:
            assert forced.tobytes() == rec.tobytes()
    # no lights: nothing to measure for shadow rays, the reference topology stays
    arrays["lights"] = arrays["lights"][:0]
    rec, entry, report = capi.choose_tree(arrays, True, 1)
    assert "-> reference topology" in report and rec.tobytes() == ref.tobytes()


def test_wider_folds_priced_on_real_queues(golden_scenes):
    """orc_nwide_stats (the analysis behind DESIGN's 8-wide decision): the SAH-optimal W-wide fold of a BVH2 walked with exact boxes.
    Wider records mean fewer record visits (and, on large scenes, more slot tests per ray) -- here: the walk terminates on the
    reference's tree and on an own one, and visits and records are monotone in W."""
    sc = golden_scenes["coverage"]
    w, h, b = 64, 48, 4
    orc = _oracle.Oracle(w, h, sc)
    orc.set_camera(T.default_camera(w, h)); orc.set_max_bounces(b)
    orc.stage("reset"); orc.stage("generate_rays")
    n = w * h
    rays = orc.buffer("rays0", T.ray, n)[: int(orc.buffer("ray_counter0", np.uint32, 1)[0])].copy()
    own = own_tree(sc["nodes"])
    for nodes in (None, own):
        prev = None
        for width in (2, 4, 8, 16):
            st = orc.nwide_stats(width, rays, False, nodes)
            assert st["rays"] == len(rays) and st["visits"] > 0 and st["records"] > 0
            if prev is not None:
                assert st["visits"] <= prev["visits"] and st["records"] <= prev["records"]
            prev = st
