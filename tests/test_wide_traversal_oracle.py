"""The 4-wide quantized traversal of the HIP path (k_trace_w4) is an algorithm of this repository, not of the reference:
oracle/oracle.c restates it ray by ray (orc_wide_trace: the kernel's own arithmetic over the records of build_wide_bvh)
and this file shows ON THE CPU that it returns what the reference's loop (trace_bvh.cl:99-211, restated as TraceOne and
pinned to the reference's kernels by tests/test_ref_pin.py) returns -- every hit record and every shadow verdict, bit for
bit, on the ray populations a path tracer actually produces: camera rays, BSDF-sampled bounces, shadow rays towards lights.
The GPU suite checks the same through the kernel itself; this check needs no GPU and also yields the walk's statistics."""
import os
import numpy as np
import pytest
from tests import _oracle
from tests.test_wide_bvh import wide_of
from raytracing_amd import host, scenes as S, types as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIGHT = ((-0.6, -1.5, 3.5), (15.0, 10.0, 5.0))


def _finish(scene, env_map):
    scene.add_directional_light(*LIGHT)
    scene.add_point_light((0.3, 0.8, 1.6), (4.0, 4.0, 3.0))
    scene.build_bvh()
    scene.set_env_image(env_map)
    scene.finalize()
    return scene.arrays()


def walk_and_compare(arrays, w, h, bounces, samples=1, aperture=0.0, collapse=1):
    """Runs the oracle stage by stage; every closest-hit and shadow queue is traced by the reference's loop AND by the
    wide walk.  Returns the wide walk's counters (closest, shadow)."""
    wide, entry = wide_of(arrays["nodes"], collapse)        # 1: SAH-optimal frontier per record (default), 2: two levels
    orc = _oracle.Oracle(w, h, arrays)
    cam = T.default_camera(w, h)
    if aperture:
        cam["aperture"] = aperture
        cam["focus_distance"] = 2.0
    orc.set_camera(cam)
    orc.set_max_bounces(bounces)
    n = w * h
    cc, cs, cd = np.zeros(10, np.uint64), np.zeros(10, np.uint64), np.zeros(10, np.uint64)
    for _ in range(samples):
        orc.stage("reset") if orc.sample_count() == 0 else None
        orc.stage("generate_rays")
        for bounce in range(bounces + 1):
            k = int(orc.buffer("ray_counter%d" % (bounce & 1), np.uint32, 1)[0])
            rays = orc.buffer("rays%d" % (bounce & 1), T.ray, n)[:k]
            orc.stage("intersect", bounce)
            want = orc.buffer("hits", T.hit, n)[:k]
            hit = want["primitive_id"] != 0xFFFFFFFF                  # bc / t are undefined for misses (trace_bvh.cl:135-136)
            for direct, cnt in ((False, cc), (True, cd)):             # the plain walk and its direct form (RT_OPT_TRACE_VARIANT 15)
                got = orc.wide_trace(wide, entry, rays, False, cnt, direct=direct)
                assert np.array_equal(got["primitive_id"], want["primitive_id"]), ("closest", bounce, direct)
                assert np.array_equal(got[hit].tobytes(), want[hit].tobytes()), ("closest", bounce, direct)
            for st, args in (("shade_miss", (bounce,)), ("clear_counters", (bounce,)), ("shade_hits", (bounce,))):
                orc.stage(st, *args)
            ks = int(orc.buffer("shadow_ray_counter", np.uint32, 1)[0])
            srays = orc.buffer("shadow_rays", T.ray, n)[:ks]
            orc.stage("intersect_shadow")
            assert np.array_equal(orc.wide_trace(wide, entry, srays, True, cs), orc.buffer("shadow_hits", np.uint32, n)[:ks]), ("shadow", bounce)
            assert np.array_equal(orc.wide_trace(wide, entry, srays, True, None, direct=True), orc.buffer("shadow_hits", np.uint32, n)[:ks])
            orc.stage("accumulate")
        orc.stage("advance")
    # the direct form visits the same nodes and leaves, with fewer trips through the stack
    for k in (0, 1, 2, 3, 4, 8, 9):
        assert cc[k] == cd[k], _oracle.Oracle.WIDE_COUNTERS[k]
    assert cd[5] <= cc[5] and cd[5] + cd[6] <= cc[5] + cc[6]
    return dict(zip(_oracle.Oracle.WIDE_COUNTERS, cc.tolist())), dict(zip(_oracle.Oracle.WIDE_COUNTERS, cs.tolist()))


def test_wide_walk_equals_the_reference_loop_on_the_golden_scenes(golden_scenes):
    for name, (w, h, b) in {"cornell": (64, 48, 5), "coverage": (72, 56, 7)}.items():
        for collapse in (1, 2):
            c, s = walk_and_compare(golden_scenes[name], w, h, b, samples=2, aperture=0.03 if name == "coverage" else 0.0, collapse=collapse)
            assert c["rays"] > 0 and s["rays"] > 0 and c["wide_visits"] > 0


def test_wide_walk_on_a_textured_city_block_and_its_statistics(env_map):
    arrays = _finish(host.Scene(arrays=S.city_block(60_000)), env_map)
    c2, s2 = walk_and_compare(arrays, 96, 54, 8, collapse=2)
    c, s = walk_and_compare(arrays, 96, 54, 8)
    # the SAH collapse is there to save visits: same rays, same leaves' triangles, fewer wide nodes on the way
    assert c["rays"] == c2["rays"] and c["triangle_tests"] == c2["triangle_tests"]
    assert c["wide_visits"] < 0.97 * c2["wide_visits"] and s["wide_visits"] < s2["wide_visits"]   # (-10 % / -7.5 % on the benchmark scene)
    # what the walk costs per ray on a Bistro-class scene (a tenth of the benchmark's triangle count): the numbers DESIGN.md quotes
    per = lambda d, k: d[k] / max(d["rays"], 1)
    assert 4 < per(c, "wide_visits") < 40 and 0.5 < per(c, "leaf_arrivals") < 20
    assert c["deepest_stack"] <= 104 and s["deepest_stack"] <= 104         # RT_W4_STACK_MAX
    assert c["rays_left_to_bvh2"] == 0


def test_wide_walk_on_a_dense_mesh_with_depth_of_field(env_map):
    tris, mats = S.cornell_blob(30_000, 3_000)
    arrays = _finish(host.Scene(arrays=dict(triangles=tris, materials=mats)), env_map)
    walk_and_compare(arrays, 80, 60, 6, aperture=0.05)


@pytest.mark.parametrize("seed", range(6))
def test_wide_walk_on_random_soups(seed, env_map):
    """Slivers, coincident triangles, lights inside the geometry, axis-parallel rays (those take the BVH2 walk, like on the GPU)."""
    rng = np.random.default_rng(seed)
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
        s.add_directional_light((0.0, 0.0, 1.0), (5.0, 5.0, 5.0))           # dir (0,0,1): 1/dir = (inf, inf, 1)
    arrays = _finish(s, env_map)
    c, sh = walk_and_compare(arrays, 48, 40, 5, collapse=1 + seed // 3)
    assert c["triangle_tests"] > 0 and sh["rays"] > 0
    if seed == 3:
        assert sh["rays_left_to_bvh2"] > 0


def test_the_pair_layout_is_a_permutation_the_walk_does_not_see(golden_scenes):
    """RT_CTX_OPT_WIDE_LAYOUT = 1 (round 6): the records in (parent, likeliest child) pairs -- a valid fold by every invariant of tests/test_wide_bvh.py, every
    record with interior slots at an even index followed by one of its children, and the CPU restatement of k_trace_w4's walk returns the reference's hits
    and verdicts on the permuted records exactly as on the fold's own order."""
    import ctypes as C
    from tests.test_wide_bvh import check, WIDE
    from raytracing_amd import capi
    lib = capi.load()
    for name, (w, h, b) in {"cornell": (48, 32, 3), "coverage": (56, 40, 4)}.items():
        arrays = golden_scenes[name]
        nodes = np.ascontiguousarray(arrays["nodes"])
        wide, entry, roots = wide_of(nodes, 1, with_roots=True)
        paired, proots = wide.copy(), roots.copy()
        assert lib.rt_debug_pair_layout(nodes.ctypes.data, len(nodes), paired.ctypes.data, proots.ctypes.data, len(paired)) == 0
        assert sorted(proots.tolist()) == sorted(roots.tolist()) and proots[0] == 0
        check(nodes, fold=(paired, entry, proots))
        interior = lambda ref: ref != 0xFFFFFFFF and not (ref & 0x80000000)
        heads = 0
        for i in range(0, len(paired) - 1, 2):
            kids = [int(r) for r in paired["ref"][i] if interior(int(r))]
            if kids:
                assert i + 1 in kids, (name, i)
                heads += 1
        with_kids = sum(1 for i in range(len(paired)) if any(interior(int(r)) for r in paired["ref"][i]))
        assert heads > 0.2 * len(paired) and heads >= 0.5 * with_kids, (heads, with_kids, len(paired))    # (most records of a 4-wide tree hold leaves only: nothing to pair them with)
        # the walk on both
        orc = _oracle.Oracle(w, h, arrays)
        orc.set_camera(T.default_camera(w, h)); orc.set_max_bounces(b)
        n = w * h
        orc.stage("reset"); orc.stage("generate_rays")
        for bounce in range(b + 1):
            k = int(orc.buffer("ray_counter%d" % (bounce & 1), np.uint32, 1)[0])
            rays = orc.buffer("rays%d" % (bounce & 1), T.ray, n)[:k]
            orc.stage("intersect", bounce)
            want = orc.buffer("hits", T.hit, n)[:k]
            hit = want["primitive_id"] != 0xFFFFFFFF
            got = orc.wide_trace(paired, entry, rays, False, None)
            assert np.array_equal(got["primitive_id"], want["primitive_id"]) and np.array_equal(got[hit].tobytes(), want[hit].tobytes())
            for st, args in (("shade_miss", (bounce,)), ("clear_counters", (bounce,)), ("shade_hits", (bounce,))):
                orc.stage(st, *args)
            ks = int(orc.buffer("shadow_ray_counter", np.uint32, 1)[0])
            srays = orc.buffer("shadow_rays", T.ray, n)[:ks]
            orc.stage("intersect_shadow")
            assert np.array_equal(orc.wide_trace(paired, entry, srays, True, None), orc.buffer("shadow_hits", np.uint32, n)[:ks])
            orc.stage("accumulate")
