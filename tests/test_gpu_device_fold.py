"""-m gpu: RT_CTX_OPT_DEVICE_FOLD (round 6) -- the collapse of a binary BVH (reference layout, src/bvh.cpp:223-245) into the 4-wide records
of k_trace_w4 on the DEVICE (raytracing_amd/csrc/fold_kernels.h) against the host's build_wide_bvh (rt_hip.hip), which tests/test_wide_bvh.py checks
invariant by invariant on the CPU: the two must agree RECORD FOR RECORD -- the same dynamic programme in the same binary64 arithmetic, the same record
roots in the same order, the same slot placement, exchange bits, frames and quantised boxes -- for the plain surface area, for the own trees' metric
and for measured per-node weights (the adaptation's folds); and a scene uploaded with the option on and off renders the same bits either way."""
import os
import numpy as np
import pytest
from tests.test_wide_bvh import WIDE, check, wide_of
from tests.test_own_tree import own_tree, wide_metric, light_dir
from tests import _oracle
from raytracing_amd import capi, host, scenes as S, types as T

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def city_nodes(n_tris):
    scene = host.Scene(arrays=S.city_block(n_tris))
    scene.build_bvh()
    return scene.arrays()["nodes"].copy()


@pytest.fixture(scope="module")
def trees(golden_scenes):
    out = {k: v["nodes"].copy() for k, v in golden_scenes.items()}
    out["city 40 K"] = city_nodes(40_000)
    out["city 300 K"] = city_nodes(300_000)
    return out


def same(dev, hst):
    (rd, ed, od), (rh, eh, oh) = dev, hst
    assert ed == eh
    assert len(rd) == len(rh), (len(rd), len(rh))
    assert np.array_equal(od, oh), "the record roots differ"
    a, b = np.ascontiguousarray(rd).view(np.uint8).reshape(len(rd), 64), np.ascontiguousarray(rh).view(np.uint8).reshape(len(rh), 64)
    bad = np.nonzero((a != b).any(axis=1))[0]
    assert len(bad) == 0, "first differing record %d of %d: device %r host %r" % (bad[0], len(a), a[bad[0]].view(WIDE), b[bad[0]].view(WIDE))


def test_the_device_fold_equals_the_host_fold_record_for_record(ctx, trees):
    for name, nodes in trees.items():
        recs, entry, roots, sec = capi.device_fold(ctx, nodes)
        h, he, hr = wide_of(nodes, 1, with_roots=True)
        same((recs, entry, roots), (h, he, hr))
        check(nodes, fold=(recs.view(WIDE).reshape(-1), entry, roots))     # ... and it is a valid fold by the CPU suite's own invariants


def test_with_the_own_trees_metric(ctx, trees):
    """the shadow rays' own tree and its projected-area metric (own_bvh.h; rt_scene_upload folds it with that metric)"""
    d = light_dir()
    for name in ("city 40 K", "coverage"):
        nodes = trees[name]
        own = own_tree(nodes, 0.5, [d])
        recs, entry, roots, _ = capi.device_fold(ctx, own, 0.5, [np.abs(d)])
        h, he = wide_metric(own, 0.5, [d])
        assert entry == he and len(recs) == len(h)
        assert np.array_equal(np.ascontiguousarray(recs).view(np.uint8).reshape(-1), np.ascontiguousarray(h).view(np.uint8).reshape(-1))


def test_with_measured_weights(ctx, trees):
    """per-node weights in place of the area: what a fold adaptation folds with (FoldAdapt: crossing counts + a surface-area prior)"""
    rng = np.random.default_rng(7)
    for name in ("city 40 K", "city 300 K", "cornell"):
        nodes = trees[name]
        w = rng.integers(0, 50, len(nodes)).astype(np.float64) + rng.random(len(nodes)) * 0.01
        dev = capi.device_fold(ctx, nodes, weights=w)
        same(dev[:3], capi.wide_bvh_weights(nodes, w))


def test_trees_that_do_not_qualify_are_refused(ctx, trees):
    nodes = trees["city 40 K"].copy()
    interior = np.nonzero((nodes["num_primitives_axis"] >> 16) == 0)[0]
    bad = nodes.copy()
    bad["bounds_max"]["x"][interior[5] + 1] += 1000.0                         # a child that sticks out of its parent
    with pytest.raises(capi.RtError):
        capi.device_fold(ctx, bad)
    with pytest.raises(capi.RtError):
        wide_of(bad)
    bad = nodes.copy()
    bad["bounds_min"]["y"][interior[3]] = np.nan
    with pytest.raises(capi.RtError):
        capi.device_fold(ctx, bad)
    leaf_root = nodes[(nodes["num_primitives_axis"] >> 16) != 0][:1].copy()   # a tree that is one leaf: no records, the entry is the leaf
    recs, entry, roots, _ = capi.device_fold(ctx, leaf_root)
    assert len(recs) == 0 and entry == (0x80000000 | int(leaf_root["offset"][0]))


def test_a_scene_renders_the_same_bits_with_the_fold_on_the_device_or_on_the_host(golden_scenes):
    """rt_scene_upload with RT_CTX_OPT_DEVICE_FOLD 1 / 0 (both trees, and the adaptation's re-folds with bit 1 | 2: waited for, small trees too)"""
    w, h, b, spp = 96, 64, 4, 6
    sc = golden_scenes["coverage"]
    cam = T.default_camera(w, h)
    images, reports = [], []
    for on in (1, 0):
        c = capi.Context(0)
        assert c.lib.rt_ctx_set_option(c.handle, 7, on) == 0
        assert c.lib.rt_ctx_set_option(c.handle, 4, 31) == 0                # adaptive fold, every bit: the first integrate waits for the re-folds
        c.upload_scene(sc)
        fr = capi.Frame(c, w, h)
        fr.set_camera(cam); fr.set_max_bounces(b)
        fr.integrate(spp)
        images.append(fr.radiance().copy())
        reports.append(c.lib.rt_scene_tree_report(c.handle).decode())
        fr.close(); c.close()
    assert "on the device" in reports[0] and "on host threads" in reports[1], reports
    assert np.array_equal(images[0], images[1], equal_nan=True)
    orc = _oracle.Oracle(w, h, sc)
    orc.set_camera(cam); orc.set_max_bounces(b); orc.integrate(spp)
    assert np.array_equal(images[0][..., :3], orc.radiance()[..., :3], equal_nan=True)


def test_one_context_takes_another_contexts_folds(golden_scenes):
    """rt_scene_export_folds / rt_scene_import_folds (one fold adaptation per process GROUP): context A adapts its folds (waited for), context B -- the same scene
    uploaded WITHOUT a shadow tree and without an adaptation of its own, as a rank > 0 of a group does -- takes A's records; both render the oracle's bits, B walks
    A's records (same bytes), and records that do not fit the scene are refused."""
    w, h, b, spp = 96, 64, 4, 5
    sc = golden_scenes["coverage"]
    cam = T.default_camera(w, h)
    A, B = capi.Context(0), capi.Context(0)
    assert A.lib.rt_ctx_set_option(A.handle, 4, 31) == 0 and A.lib.rt_ctx_set_option(A.handle, 2, 2) == 0     # adapt + wait, own shadow tree forced
    assert B.lib.rt_ctx_set_option(B.handle, 4, 0) == 0 and B.lib.rt_ctx_set_option(B.handle, 2, 0) == 0
    A.upload_scene(sc); B.upload_scene(sc)
    fa, fb = capi.Frame(A, w, h), capi.Frame(B, w, h)
    for f in (fa, fb):
        f.set_camera(cam); f.set_max_bounces(b)
    fa.integrate(2)                                                           # A's first integrate probes, folds again and adopts
    cl, sh, ent = capi.export_folds(A.handle)
    assert len(cl) > 0 and len(sh) > 0
    fb.integrate(2)
    capi.import_folds(B.handle, cl, sh, ent)
    assert "imported folds" in B.lib.rt_scene_tree_report(B.handle).decode()
    cl2, sh2, ent2 = capi.export_folds(B.handle)
    assert ent2 == ent and np.array_equal(cl2, cl) and np.array_equal(sh2, sh)
    fa.integrate(spp - 2); fb.integrate(spp - 2)
    assert np.array_equal(fa.radiance(), fb.radiance(), equal_nan=True)
    orc = _oracle.Oracle(w, h, sc)
    orc.set_camera(cam); orc.set_max_bounces(b); orc.integrate(spp)
    assert np.array_equal(fb.radiance()[..., :3], orc.radiance()[..., :3], equal_nan=True)
    # refused: a ref outside the records, a leaf outside the triangles
    bad = cl.copy().view(np.uint32).reshape(len(cl), 16)
    bad[0, 10] = len(cl) + 5                                                  # ref[0] of record 0
    with pytest.raises(capi.RtError, match="do not fit"):
        capi.import_folds(B.handle, bad.view(np.uint8).reshape(len(cl), 64), sh, ent)
    fb.integrate(1); fa.integrate(1)                                           # ... and B is still intact
    assert np.array_equal(fa.radiance(), fb.radiance(), equal_nan=True)
    fa.close(); fb.close(); A.close(); B.close()


def test_the_tree_built_on_the_device_is_a_tree_over_the_references_leaves(ctx, trees, golden_scenes):
    """RT_CTX_OPT_TREE_BUILDER = 1 (ploc_kernels.h): PLOC on the device over the reference's leaves.  Structure by the CPU suite's own checker (tests/test_own_tree.py:
    exactly the reference's leaves -- first triangle, count, exact box -- in the reference's linear layout, interior boxes exact unions), the same tree every run, and
    the CPU restatement of k_trace_w4's walk over its fold returns the reference's shadow verdicts."""
    from tests.test_own_tree import check_own_structure, shadow_and_closest_on
    d = light_dir()
    for name in ("cornell", "coverage", "city 40 K", "city 300 K"):
        nodes = trees[name]
        own, sec, rounds = capi.device_tree(ctx, nodes, 0.5, [d])
        check_own_structure(nodes, own)
        again, _, _ = capi.device_tree(ctx, nodes, 0.5, [d])
        assert np.array_equal(own.view(np.uint8), again.view(np.uint8)), name      # no choice depends on which thread arrives first
        assert rounds < 200, (name, rounds)
        recs, entry, roots, _ = capi.device_fold(ctx, own, 0.5, [np.abs(d)])
        h, he = wide_metric(own, 0.5, [d])
        assert entry == he and np.array_equal(np.ascontiguousarray(recs).view(np.uint8).reshape(-1), np.ascontiguousarray(h).view(np.uint8).reshape(-1))
    for name, (w, h, b) in {"cornell": (48, 32, 3), "coverage": (56, 40, 4)}.items():
        arrays = golden_scenes[name]
        own, _, _ = capi.device_tree(ctx, arrays["nodes"], 0.5, [d])
        shadow_and_closest_on(arrays, {"device": wide_metric(own, 0.5, [d])}, w, h, b)    # asserts equal shadow verdicts at every bounce


def test_the_device_built_tree_is_about_as_good_as_the_host_built_one(ctx, trees):
    """quality, by the library's own yardstick: the metric summed over the interior boxes (what both builders minimise greedily)"""
    d = light_dir()
    for name in ("city 40 K", "city 300 K"):
        nodes = trees[name]
        host_tree = own_tree(nodes, 0.5, [d])
        dev_tree, sec, rounds = capi.device_tree(ctx, nodes, 0.5, [d])
        def cost(t):
            inner = (t["num_primitives_axis"] >> 16) == 0
            e = np.stack([t["bounds_max"][c].astype(np.float64) - t["bounds_min"][c].astype(np.float64) for c in "xyz"], 1)[inner]
            iso = 0.5 * 0.5 * (e[:, 0] * e[:, 1] + e[:, 1] * e[:, 2] + e[:, 2] * e[:, 0])
            proj = abs(d[0]) * e[:, 1] * e[:, 2] + abs(d[1]) * e[:, 2] * e[:, 0] + abs(d[2]) * e[:, 0] * e[:, 1]
            return float((iso + proj).sum())
        ratio = cost(dev_tree) / cost(host_tree)
        print("%s: device-built / host-built interior cost %.3f, %d rounds, %.3f s" % (name, ratio, rounds, sec))
        assert ratio < 1.35, (name, ratio)


def test_a_scene_with_its_shadow_tree_built_on_the_device_renders_the_references_bits(golden_scenes):
    w, h, b, spp = 96, 64, 4, 4
    sc = golden_scenes["coverage"]
    cam = T.default_camera(w, h)
    c = capi.Context(0)
    assert c.lib.rt_ctx_set_option(c.handle, 9, 1) == 0 and c.lib.rt_ctx_set_option(c.handle, 2, 2) == 0      # device builder, own shadow tree forced
    assert c.lib.rt_ctx_set_option(c.handle, 4, 31) == 0                                                     # ... and adapted (rotated, re-folded) before the frame's rays
    c.upload_scene(sc)
    rep = c.lib.rt_scene_tree_report(c.handle).decode()
    assert "on the device (PLOC)" in rep, rep
    fr = capi.Frame(c, w, h)
    fr.set_camera(cam); fr.set_max_bounces(b)
    fr.integrate(spp)
    orc = _oracle.Oracle(w, h, sc)
    orc.set_camera(cam); orc.set_max_bounces(b); orc.integrate(spp)
    assert np.array_equal(fr.radiance()[..., :3], orc.radiance()[..., :3], equal_nan=True)
    st = fr.stats()
    assert (st.closest_rays, st.shadow_rays) == orc.ray_totals()
    fr.close(); c.close()
