"""RT_CTX_OPT_ADAPTIVE_FOLD, host half (no GPU): the 4-wide fold of the reference's BVH2 made again for MEASURED box-pass frequencies of
a frame's own rays (rt_hip.hip: FoldAdapt, refold_for_rays; exported for tests as rt_debug_adapt_fold).

What has to hold:
 * an adapted fold is a valid fold -- every record a frontier of its root, every order table the reference's visit order, every box
   conservative on the exact grid (tests/test_wide_bvh.py: check), every leaf of the reference once;
 * k_trace_w4's walk restated on the CPU (oracle.c: orc_wide_trace) over the adapted records returns the reference loop's hits and shadow
   verdicts BIT FOR BIT, on rays the fold was not adapted to as well;
 * and it is cheaper for rays it has not seen (out of sample), which is the point (tools/fold_weight_study.py has the numbers at scale).
The GPU half (probe frame, hand-over between two rt_integrate calls) is tests/test_gpu_parity.py::test_adaptive_fold_*."""
import numpy as np
import pytest
from tests import _oracle
from tests.test_wide_bvh import WIDE, check, wide_of
from tests.test_own_tree import _finish
from raytracing_amd import capi, host, scenes as S, types as T


def queues_of(arrays, w, h, bounces):
    """Oracle stage by stage: per bounce the closest-hit queue, its hits, the shadow queue, its verdicts."""
    orc = _oracle.Oracle(w, h, arrays)
    orc.set_camera(T.default_camera(w, h))
    orc.set_max_bounces(bounces)
    n = w * h
    out = []
    orc.stage("reset")
    orc.stage("generate_rays")
    for bounce in range(bounces + 1):
        k = int(orc.buffer("ray_counter%d" % (bounce & 1), np.uint32, 1)[0])
        rays = orc.buffer("rays%d" % (bounce & 1), T.ray, n)[:k].copy()
        orc.stage("intersect", bounce)
        hits = orc.buffer("hits", T.hit, n)[:k].copy()
        for st, args in (("shade_miss", (bounce,)), ("clear_counters", (bounce,)), ("shade_hits", (bounce,))):
            orc.stage(st, *args)
        ks = int(orc.buffer("shadow_ray_counter", np.uint32, 1)[0])
        srays = orc.buffer("shadow_rays", T.ray, n)[:ks].copy()
        orc.stage("intersect_shadow")
        verdicts = orc.buffer("shadow_hits", np.uint32, n)[:ks].copy()
        orc.stage("accumulate")
        out.append((rays, hits, srays, verdicts))
    return orc, out


def as_probe(rays, hits=None):
    """what fold_probe hands the worker: origin + t_max (the hit distance where there was one), direction"""
    o = np.stack([rays["origin"][c] for c in "xyz"] + [rays["direction"]["w"]], 1).astype(np.float32)
    d = np.stack([rays["direction"][c] for c in "xyz"] + [np.zeros(len(rays), np.float32)], 1).astype(np.float32)
    if hits is not None:
        hit = (hits["primitive_id"] != 0xFFFFFFFF) & (hits["t"] > 0) & (hits["t"] * np.float32(1.0001) < o[:, 3])
        o[hit, 3] = hits["t"][hit] * np.float32(1.0001)
    return o, d


def same_hits(got, want):
    hit = want["primitive_id"] != 0xFFFFFFFF
    bad = got["primitive_id"] != want["primitive_id"]
    bad |= hit & (got["t"] != want["t"])
    bad |= hit & (np.ascontiguousarray(got["bc"]).view(np.float32).reshape(-1, 2) != np.ascontiguousarray(want["bc"]).view(np.float32).reshape(-1, 2)).any(1)
    return int(bad.sum())


def adapted_and_checked(nodes, o, d):
    rec, entry, roots, cost, adopted = capi.adapt_fold(nodes, o, d)
    wide = rec.view(WIDE).reshape(-1)
    check(nodes, 1, fold=(wide, entry, roots))
    return wide, entry, cost, adopted


def exercise(arrays, w, h, bounces):
    nodes = arrays["nodes"]
    orc, q = queues_of(arrays, w, h, bounces)
    ref_wide, ref_entry = wide_of(nodes, 1)
    # adapted to every second ray, walked by all of them (in sample and out of sample alike: results may not depend on it)
    o = np.concatenate([as_probe(r[::2], hh[::2])[0] for r, hh, _, _ in q]); d = np.concatenate([as_probe(r[::2])[1] for r, _, _, _ in q])
    so = np.concatenate([as_probe(s[::2])[0] for _, _, s, _ in q]); sdir = np.concatenate([as_probe(s[::2])[1] for _, _, s, _ in q])
    cl_wide, cl_entry, cl_cost, cl_adopted = adapted_and_checked(nodes, o, d)
    sh_wide, sh_entry, sh_cost, sh_adopted = adapted_and_checked(nodes, so, sdir)
    visits = {k: np.zeros(10, np.uint64) for k in ("ref closest", "adapted closest", "ref shadow", "adapted shadow")}
    for rays, hits, srays, verdicts in q:
        assert same_hits(orc.wide_trace(cl_wide, cl_entry, rays, False, None, direct=True), hits) == 0
        assert same_hits(orc.wide_trace(sh_wide, sh_entry, rays, False, None, direct=True), hits) == 0     # ... and a fold made for other rays
        for direct in (False, True):
            assert np.array_equal(orc.wide_trace(sh_wide, sh_entry, srays, True, None, direct=direct), verdicts)
            assert np.array_equal(orc.wide_trace(cl_wide, cl_entry, srays, True, None, direct=direct), verdicts)
        # what the unseen rays (the odd ones) pay
        orc.wide_trace(ref_wide, ref_entry, rays[1::2], False, visits["ref closest"], direct=True)
        orc.wide_trace(cl_wide, cl_entry, rays[1::2], False, visits["adapted closest"], direct=True)
        orc.wide_trace(ref_wide, ref_entry, srays[1::2], True, visits["ref shadow"], direct=True)
        orc.wide_trace(sh_wide, sh_entry, srays[1::2], True, visits["adapted shadow"], direct=True)
    return visits, (cl_cost, cl_adopted), (sh_cost, sh_adopted)


def test_adapted_folds_on_the_golden_scenes(golden_scenes):
    for name, arrays in golden_scenes.items():
        visits, (cl_cost, _), (sh_cost, _) = exercise(arrays, 24, 16, 3)
        # optimal for its own weights: never costlier than the surface-area fold on them
        assert cl_cost[1] <= cl_cost[0] * (1 + 1e-12) and sh_cost[1] <= sh_cost[0] * (1 + 1e-12), name


def test_adapted_fold_of_a_city_block_is_cheaper_for_rays_it_has_not_seen(env_map):
    scene = host.Scene(arrays=S.city_block(20000))
    arrays = _finish(scene, env_map, point=False)
    visits, (cl_cost, cl_adopted), (sh_cost, sh_adopted) = exercise(arrays, 96, 54, 4)
    node_visits = lambda k: int(visits[k][_oracle.Oracle.WIDE_COUNTERS.index("wide_visits")])
    assert cl_adopted and sh_adopted
    assert cl_cost[1] < 0.97 * cl_cost[0] and sh_cost[1] < 0.97 * sh_cost[0]
    # out of sample, on the walk the kernel makes (quantised boxes, shrinking t_max)
    assert node_visits("adapted closest") < 0.985 * node_visits("ref closest"), (node_visits("adapted closest"), node_visits("ref closest"))
    assert node_visits("adapted shadow") < 0.985 * node_visits("ref shadow"), (node_visits("adapted shadow"), node_visits("ref shadow"))


@pytest.mark.parametrize("seed", range(4))
def test_adapted_folds_of_random_soups(seed, env_map):
    """Slivers and coincident triangles (ties): the hit must still be the reference's, whatever the fold."""
    rng = np.random.default_rng(500 + seed)
    n = int(rng.integers(40, 900))
    P = rng.normal(size=(n, 1, 3)) * 1.2 + rng.normal(size=(n, 3, 3)) * float(10.0 ** rng.uniform(-1.2, 0.0)) + np.array([0.0, 2.5, 1.0])
    P = P.astype(np.float32)
    if seed % 2:
        P[: n // 4] = P[0]
    N = np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 0])
    N = (N / np.maximum(np.linalg.norm(N, axis=1, keepdims=True), 1e-20)).astype(np.float32)[:, None, :].repeat(3, 1)
    tris = S.to_triangles([(P, N, np.zeros((n, 3, 2), np.float32), 0)])
    mats = np.array([S.make_material(kd=(0.7, 0.6, 0.5), ks=(0.3, 0.3, 0.3), roughness=0.3)], dtype=T.packed_material)
    arrays = _finish(host.Scene(arrays=dict(triangles=tris, materials=mats)), env_map)
    if (arrays["nodes"]["num_primitives_axis"][0] >> 16) != 0:
        pytest.skip("a single leaf: no tree to fold")
    exercise(arrays, 40, 30, 4)


def test_adapted_fold_of_the_shadow_rays_own_tree(env_map):
    """The shadow rays' own binary tree (own_bvh.h) is what FoldAdapt folds again when rt_scene_upload chose it: verdicts stay the reference
    loop's, and the adapted fold of the own tree is cheaper than its projected-area fold for rays it has not seen."""
    from tests.test_own_tree import own_tree, wide_metric, check_own_structure, light_dir
    scene = host.Scene(arrays=S.city_block(20000))
    arrays = _finish(scene, env_map, point=False)
    nodes = arrays["nodes"]
    d = light_dir()
    own = own_tree(nodes, 0.5, [d])
    check_own_structure(nodes, own)
    area_wide, area_entry = wide_metric(own, 0.5, [d])
    orc, q = queues_of(arrays, 96, 54, 4)
    so = np.concatenate([as_probe(s[::2])[0] for _, _, s, _ in q]); sdir = np.concatenate([as_probe(s[::2])[1] for _, _, s, _ in q])
    rec, entry, roots, cost, adopted = capi.adapt_fold(own, so, sdir)
    wide = rec.view(WIDE).reshape(-1)
    check(own, 1, fold=(wide, entry, roots))
    assert adopted and cost[1] < cost[0]
    v_area, v_adapted = np.zeros(10, np.uint64), np.zeros(10, np.uint64)
    for rays, hits, srays, verdicts in q:
        for direct in (False, True):
            assert np.array_equal(orc.wide_trace(wide, entry, srays, True, None, direct=direct), verdicts)
        orc.wide_trace(area_wide, area_entry, srays[1::2], True, v_area, direct=True)
        orc.wide_trace(wide, entry, srays[1::2], True, v_adapted, direct=True)
    assert int(v_adapted[1]) < int(v_area[1]), (int(v_adapted[1]), int(v_area[1]))


def rotated_and_checked(arrays, w, h, bounces):
    """tree_rotate.h (RT_CTX_OPT_ADAPTIVE_FOLD bit 3): the reference's tree and the backend's own, each rotated for every second shadow ray's
    crossings, are still binary trees over exactly the reference's leaves with exact-union boxes; their adapted folds are valid folds; and the
    walk over them returns the reference loop's verdict for every shadow ray.  Returns the wide visits of the unseen rays per tree."""
    from tests.test_own_tree import own_tree, check_own_structure, light_dir
    nodes = arrays["nodes"]
    orc, q = queues_of(arrays, w, h, bounces)
    so = np.concatenate([as_probe(s[::2])[0] for _, _, s, _ in q]); sdir = np.concatenate([as_probe(s[::2])[1] for _, _, s, _ in q])
    if len(so) == 0:
        return None
    visits = {}
    for name, tree in (("reference", nodes), ("own", own_tree(nodes, 0.5, [light_dir()]))):
        rot, crossings, made = capi.rotate_tree(tree, so, sdir, 8)
        check_own_structure(nodes, rot)
        assert crossings[1] <= crossings[0] and (made == 0) == (crossings[1] == crossings[0])
        for label, t in ((name, tree), (name + " rotated", rot)):
            rec, entry, roots, cost, adopted = capi.adapt_fold(t, so, sdir)
            wide = rec.view(WIDE).reshape(-1)
            check(t, 1, fold=(wide, entry, roots))
            v = np.zeros(10, np.uint64)
            for rays, hits, srays, verdicts in q:
                for direct in (False, True):
                    assert np.array_equal(orc.wide_trace(wide, entry, srays, True, None, direct=direct), verdicts), (label, direct)
                orc.wide_trace(wide, entry, srays[1::2], True, v, direct=True)
            visits[label] = int(v[1])
    return visits


def test_rotated_shadow_trees_on_the_golden_scenes(golden_scenes):
    for name, arrays in golden_scenes.items():
        rotated_and_checked(arrays, 24, 16, 3)


def test_rotated_shadow_tree_of_a_city_block_is_cheaper_for_rays_it_has_not_seen(env_map):
    scene = host.Scene(arrays=S.city_block(20000))
    arrays = _finish(scene, env_map, point=False)
    v = rotated_and_checked(arrays, 96, 54, 4)
    assert v["own rotated"] < 0.97 * v["own"] and v["reference rotated"] < 0.97 * v["reference"], v


@pytest.mark.parametrize("seed", range(3))
def test_rotated_shadow_trees_of_random_soups(seed, env_map):
    rng = np.random.default_rng(700 + seed)
    n = int(rng.integers(60, 700))
    P = rng.normal(size=(n, 1, 3)) * 1.2 + rng.normal(size=(n, 3, 3)) * float(10.0 ** rng.uniform(-1.2, 0.0)) + np.array([0.0, 2.5, 1.0])
    P = P.astype(np.float32)
    if seed % 2:
        P[: n // 4] = P[0]
    N = np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 0])
    N = (N / np.maximum(np.linalg.norm(N, axis=1, keepdims=True), 1e-20)).astype(np.float32)[:, None, :].repeat(3, 1)
    tris = S.to_triangles([(P, N, np.zeros((n, 3, 2), np.float32), 0)])
    mats = np.array([S.make_material(kd=(0.7, 0.6, 0.5), ks=(0.3, 0.3, 0.3), roughness=0.3)], dtype=T.packed_material)
    arrays = _finish(host.Scene(arrays=dict(triangles=tris, materials=mats)), env_map)
    if (arrays["nodes"]["num_primitives_axis"][0] >> 16) != 0:
        pytest.skip("a single leaf: no tree to rotate")
    rotated_and_checked(arrays, 40, 30, 4)


def test_the_workers_shadow_side_picks_the_cheaper_candidate(env_map):
    """adapt_shadow_side (rt_hip.hip) as the worker thread runs it: without bit 3 the fold of the tree as it is; with it the rotated tree's fold
    when that costs the probe rays less -- and whichever it hands over is a valid fold of the tree it names, with the reference's verdicts."""
    from tests.test_own_tree import check_own_structure
    scene = host.Scene(arrays=S.city_block(20000))
    arrays = _finish(scene, env_map, point=False)
    nodes = arrays["nodes"]
    orc, q = queues_of(arrays, 96, 54, 4)
    so = np.concatenate([as_probe(s[::2])[0] for _, _, s, _ in q]); sdir = np.concatenate([as_probe(s[::2])[1] for _, _, s, _ in q])
    got = {}
    for mode in (5, 13):
        rec, entry, roots, tree, cost, made, adopted, moved = capi.adapt_shadow_side(nodes, so, sdir, mode)
        wide = rec.view(WIDE).reshape(-1)
        assert adopted and cost[1] < cost[0] and moved == 0
        assert (made != 0) == bool(mode & 8)
        if made:
            check_own_structure(nodes, tree)
        else:
            assert np.array_equal(tree, nodes)
        check(tree, 1, fold=(wide, entry, roots))
        for rays, hits, srays, verdicts in q:
            assert np.array_equal(orc.wide_trace(wide, entry, srays, True, None, direct=True), verdicts)
        got[mode] = cost
    assert got[13][0] == got[5][0] and got[13][1] < got[5][1]        # the same current fold; the rotated tree's fold is the cheaper candidate


def test_occluder_first_slot_order_is_a_permutation_that_changes_no_verdict(env_map):
    """Bit 4: k_trace_w4<shadow> looks at a record's slots in stored order and an any-hit verdict is an OR, so the worker may store them likeliest
    occluder first (by the probe rays' nearest occluders, found on the host).  Per record a pure permutation of (ref, box bytes); the walk returns
    the reference loop's verdict for every ray; occluded rays it has not seen take fewer steps."""
    scene = host.Scene(arrays=S.city_block(20000))
    arrays = _finish(scene, env_map, point=False)
    nodes = arrays["nodes"]
    orc, q = queues_of(arrays, 96, 54, 4)
    so = np.concatenate([as_probe(s[::2])[0] for _, _, s, _ in q]); sdir = np.concatenate([as_probe(s[::2])[1] for _, _, s, _ in q])

    def slots(wide):
        """per record the set of (ref, lo bytes, hi bytes) of its four slots"""
        lo = np.ascontiguousarray(wide["lo"]).view(np.uint8).reshape(len(wide), 3, 4)
        hi = np.ascontiguousarray(wide["hi"]).view(np.uint8).reshape(len(wide), 3, 4)
        return [sorted((int(wide["ref"][w][k]), lo[w, :, k].tobytes(), hi[w, :, k].tobytes()) for k in range(4)) for w in range(len(wide))]

    for base_mode in (5, 13):
        plain = capi.adapt_shadow_side(nodes, so, sdir, base_mode, arrays["triangles"])
        first = capi.adapt_shadow_side(nodes, so, sdir, base_mode | 16, arrays["triangles"])
        w0, w1 = plain[0].view(WIDE).reshape(-1), first[0].view(WIDE).reshape(-1)
        assert plain[7] == 0 and first[7] > 0 and first[1] == plain[1] and np.array_equal(first[2], plain[2]) and np.array_equal(first[3], plain[3])
        assert slots(w0) == slots(w1)                                   # the same slots in every record ...
        assert int((w0["ref"] != w1["ref"]).any(1).sum()) == first[7]   # ... in another order in exactly the records it reports
        for f in ("origin", "meta"):
            assert np.array_equal(w0[f], w1[f])
        steps = {}
        for name, wide in (("as placed", w0), ("occluder first", w1)):
            v = np.zeros(10, np.uint64)
            for rays, hits, srays, verdicts in q:
                for direct in (False, True):
                    assert np.array_equal(orc.wide_trace(wide, first[1], srays, True, None, direct=direct), verdicts), (name, direct)
                unseen = srays[1::2]; occluded = verdicts[1::2] != 0xFFFFFFFF
                orc.wide_trace(wide, first[1], unseen[occluded], True, v, direct=True)
            steps[name] = int(v[1] + v[4])
        assert steps["occluder first"] < 0.97 * steps["as placed"], steps


def test_an_adaptation_in_flight_is_abandoned_not_waited_for(env_map):
    """rt_scene_upload / rt_ctx_destroy while the worker thread is counting, rotating or folding: the FoldAdapt is dropped, its worker sees the flag at
    its next check and the drop returns -- in a small fraction of what the adaptation itself takes, whenever it comes."""
    import ctypes as C
    lib = capi.load()
    scene = host.Scene(arrays=S.city_block(150_000))
    arrays = _finish(scene, env_map, point=False)
    nodes = np.ascontiguousarray(arrays["nodes"])
    orc, q = queues_of(arrays, 160, 90, 4)
    o = np.concatenate([as_probe(r, hh)[0] for r, hh, _, _ in q]); d = np.concatenate([as_probe(r)[1] for r, _, _, _ in q])
    o, d = np.ascontiguousarray(o), np.ascontiguousarray(d)
    done = C.c_int()
    full = lib.rt_debug_fold_abandon(nodes.ctypes.data, len(nodes), o.ctypes.data, d.ctypes.data, len(o), 13, 4000, C.byref(done))
    assert full >= 0 and done.value == 1                                # left alone, it finishes (well within 4 s) and the drop is a join of nothing
    for delay in (0, 3, 15, 60):
        ms = lib.rt_debug_fold_abandon(nodes.ctypes.data, len(nodes), o.ctypes.data, d.ctypes.data, len(o), 13, delay, C.byref(done))
        assert 0 <= ms < 2000, (delay, ms)                              # (generous: a loaded CI host; the checks are a few milliseconds apart)


def test_rays_that_pass_nothing_leave_the_fold_alone(golden_scenes):
    arrays = next(iter(golden_scenes.values()))
    o = np.array([[1e6, 1e6, 1e6, 1.0]], np.float32); d = np.array([[1.0, 0.0, 0.0, 0.0]], np.float32)
    with pytest.raises(capi.RtError):
        capi.adapt_fold(arrays["nodes"], o, d)


def test_rotation_refuses_a_node_array_that_is_not_a_tree(golden_scenes):
    """rt_debug_rotate_tree is an exported entry point and tree_rotate.h's rotate() used to check child indices only locally -- two
    interior nodes sharing a child (a DAG) were walked once per path and caught only by the size check at the very end.  Now every node but the
    root must be the child of exactly one interior node, checked before anything is walked."""
    arrays = golden_scenes["coverage"]
    nodes = arrays["nodes"].copy()
    orc, q = queues_of(arrays, 24, 16, 2)
    so = np.concatenate([as_probe(s)[0] for _, _, s, _ in q]); sdir = np.concatenate([as_probe(s)[1] for _, _, s, _ in q])
    assert len(so) > 0
    rot, crossings, made = capi.rotate_tree(nodes, so, sdir, 2)             # the tree as it is: accepted
    assert len(rot) == len(nodes)
    interior = [i for i in range(len(nodes)) if (int(nodes[i]["num_primitives_axis"]) >> 16) == 0]
    a = next(i for i in interior if (int(nodes[i + 1]["num_primitives_axis"]) >> 16) == 0 and int(nodes[i + 1]["offset"]) > i + 2)
    bad = nodes.copy()
    bad[a]["offset"] = int(nodes[a + 1]["offset"])                         # a's second child is now also its first child's second child
    with pytest.raises(capi.RtError, match="not a tree"):
        capi.rotate_tree(bad, so, sdir, 2)


def test_when_the_camera_has_left_the_view():
    """fold_view_left (rt_hip.hip): the thresholds that arm a new adaptation -- 3 % of the scene's diagonal, 20 degrees, a tenth of the field of view"""
    lib = capi.load()
    base = T.default_camera(64, 64)

    def left(cam, diagonal=100.0):
        a, b = np.ascontiguousarray(base), np.ascontiguousarray(cam)
        return lib.rt_debug_fold_view_left(a.ctypes.data, b.ctypes.data, diagonal)

    assert left(base) == 0
    for dx, want in ((2.9, 0), (3.1, 1)):                           # 3 % of a diagonal of 100
        cam = base.copy(); cam["position"]["x"] = float(base["position"]["x"]) + dx
        assert left(cam) == want, dx
    cam = base.copy(); cam["position"]["z"] = float(base["position"]["z"]) + 2.0
    assert left(cam, 100.0) == 0 and left(cam, 50.0) == 1           # the same step in a smaller scene
    f = np.array([float(base["front"][k]) for k in "xyz"])
    side = np.cross(f, [0.0, 0.0, 1.0]); side /= np.linalg.norm(side)
    for deg, want in ((19.0, 0), (21.0, 1), (180.0, 1)):
        g = np.cos(np.radians(deg)) * f + np.sin(np.radians(deg)) * side
        cam = base.copy()
        for i, k in enumerate("xyz"):
            cam["front"][k] = g[i]
        assert left(cam) == want, deg
    for scale, want in ((1.09, 0), (1.11, 1), (0.89, 1)):
        cam = base.copy(); cam["fov"] = np.float32(float(base["fov"]) * scale)
        assert left(cam) == want, scale
    cam = base.copy(); cam["position"]["x"] = np.float32(np.nan)
    assert left(cam) in (0, 1)                                      # no crash on garbage; whatever it answers, results do not depend on it
    assert lib.rt_debug_fold_view_left(None, None, 1.0) == -1
