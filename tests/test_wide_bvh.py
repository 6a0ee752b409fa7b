"""The 4-wide quantized tree of k_trace_w4 (build_wide_bvh in rt_hip.hip), checked on the host:
the invariants the exactness argument of DESIGN.md rests on.

  * every slot box, dequantised EXACTLY in binary32 the way the kernel does it
    (fma(q, cell, origin)), contains the reference node's box;
  * origin + q * cell is exactly representable for q = 0..255 (no rounding in the dequantisation);
  * the slots of a record are a frontier of the BVH2 subtree it folds (at most three interior nodes opened), in the BVH2's
    depth-first order; the order table brings them into the reference's visit order (near child first at every folded
    node, trace_bvh.cl:181-190) for each of the eight direction octants; the leaves reachable from the wide tree are
    exactly the reference's leaves, each once;
  * the SAH collapse is optimal for its cost (sum of the folded nodes' areas), by exhaustive search on small trees;
  * trees that do not qualify (bounds not nested / not finite) are refused."""
import ctypes as C
import numpy as np
import pytest
from raytracing_amd import capi, host, scenes as S, types as T

LEAF, EMPTY = 0x80000000, 0xFFFFFFFF
WIDE = np.dtype([("origin", "<f4", 3), ("meta", "<u4"), ("lo", "<u4", 3), ("hi", "<u4", 3), ("ref", "<u4", 4), ("order", "<u4"), ("pad", "<u4")])
assert WIDE.itemsize == 64


def wide_of(nodes, collapse=1, with_roots=False):
    """collapse: 1 = the SAH-optimal frontier per record (the default of rt_scene_upload), 2 = two BVH2 levels per record"""
    lib = capi.load()
    n, entry = C.c_uint32(), C.c_uint32()
    nodes = np.ascontiguousarray(nodes)
    rc = lib.rt_debug_wide_bvh(nodes.ctypes.data, len(nodes), collapse, None, None, 0, C.byref(n), C.byref(entry))
    if rc != 0:
        raise capi.RtError(lib.rt_last_error(None).decode())
    out = np.zeros(n.value, WIDE)
    roots = np.zeros(n.value, np.uint32)
    assert lib.rt_debug_wide_bvh(nodes.ctypes.data, len(nodes), collapse, out.ctypes.data, roots.ctypes.data, len(out), C.byref(n), C.byref(entry)) == 0
    return (out, entry.value, roots) if with_roots else (out, entry.value)


def bvh_of(tris, mats):
    s = host.Scene(arrays=dict(triangles=tris, materials=mats))
    s.build_bvh()
    a = s.arrays()
    return a["nodes"].copy(), a["triangles"].copy()


NETWORK = ((0, 1), (2, 3), (0, 2), (1, 3))              # the four conditional exchanges of w4_test_slots, in its order


def area(nodes, n):
    d = [float(nodes["bounds_max"][c][n]) - float(nodes["bounds_min"][c][n]) for c in "xyz"]
    return d[0] * d[1] + d[1] * d[2] + d[2] * d[0]


def check(nodes, collapse=1, fold=None):
    """fold: (records, entry, roots) of another fold of `nodes` to validate instead (the adapted folds of tests/test_adaptive_fold.py)"""
    wide, entry, roots = fold if fold is not None else wide_of(nodes, collapse, with_roots=True)
    is_leaf = (nodes["num_primitives_axis"] >> 16) != 0
    if is_leaf[0]:
        assert len(wide) == 0 and entry == (LEAF | int(nodes["offset"][0]))
        return wide
    assert entry == 0 and roots[0] == 0
    record_of = {int(n): w for w, n in enumerate(roots)}
    assert len(record_of) == len(wide)
    bmin = np.stack([nodes["bounds_min"][c] for c in "xyz"], 1)
    bmax = np.stack([nodes["bounds_max"][c] for c in "xyz"], 1)
    kids = lambda n: (n + 1, int(nodes["offset"][n]))
    seen_leaves = []
    f32 = np.float32
    for w, rec in enumerate(wide):
        n = int(roots[w])
        assert not is_leaf[n]
        meta = int(rec["meta"])
        cell = [f32(2.0) ** f32(((meta >> (8 * a)) & 0xFF) - 127) for a in range(3)]
        refs = [int(x) for x in rec["ref"]]
        n_slots = meta >> 24
        occupied = [k for k in range(4) if refs[k] != EMPTY]
        assert 2 <= n_slots <= 4 and len(occupied) == n_slots
        # the occupied slots are a frontier of n's subtree (stored wherever the exchange network needs them): open n, then
        # whichever frontier node is not a slot, at most three in all
        def slot_of(c):
            want = (LEAF | int(nodes["offset"][c])) if is_leaf[c] else record_of.get(c)
            return refs.index(want) if want in refs else None
        frontier, opened = [n], set()
        while True:
            at = next((i for i, c in enumerate(frontier) if c == n or slot_of(c) is None), None)
            if at is None:
                break
            c = frontier[at]
            assert not is_leaf[c] and len(opened) < 3, (w, frontier, refs)
            opened.add(c)
            frontier[at:at + 1] = kids(c)
            if c == n:
                n = -1                                                # the root is opened exactly once
        n = int(roots[w])
        assert len(frontier) == n_slots and sorted(slot_of(c) for c in frontier) == occupied
        if collapse == 2:                                             # two levels: both children, and theirs
            assert opened == {n} | {c for c in kids(n) if not is_leaf[c]}
        # the order table: after the exchanges it asks for, the occupied slots stand in the reference's visit order
        def visit(c, octant):
            if c not in opened:
                return [slot_of(c)]
            first, second = kids(c)
            if (octant >> (int(nodes["num_primitives_axis"][c]) & 0xFFFF)) & 1:
                first, second = second, first
            return visit(first, octant) + visit(second, octant)
        for o in range(8):
            bits = (int(rec["order"]) >> (4 * o)) & 0xF
            pos = [0, 1, 2, 3]
            for b, (i, j) in enumerate(NETWORK):
                if (bits >> b) & 1:
                    pos[i], pos[j] = pos[j], pos[i]
            assert [k for k in pos if k in occupied] == visit(n, o), (w, o, pos, visit(n, o))
        frontier = sorted(frontier, key=slot_of)
        for child in frontier:
            k = slot_of(child)
            for a in range(3):
                qlo, qhi = (int(rec["lo"][a]) >> (8 * k)) & 0xFF, (int(rec["hi"][a]) >> (8 * k)) & 0xFF
                o = f32(rec["origin"][a])
                # exact representability: the fp32 product and sum carry no rounding error
                lo32 = f32(f32(qlo) * cell[a]) + o
                hi32 = f32(f32(qhi) * cell[a]) + o
                assert float(lo32) == float(o) + qlo * float(cell[a])
                assert float(hi32) == float(o) + qhi * float(cell[a])
                assert float(o) / float(cell[a]) == round(float(o) / float(cell[a]))
                assert abs(float(o) / float(cell[a])) + 255 < 2 ** 24
                # conservative: the stored box contains the reference's box
                assert lo32 <= bmin[child, a] and hi32 >= bmax[child, a], (w, k, a)
                # and is tight to the grid
                assert float(bmin[child, a]) - float(lo32) < float(cell[a]) and float(hi32) - float(bmax[child, a]) < float(cell[a])
            if is_leaf[child]:
                seen_leaves.append(child)
        for k in set(range(4)) - set(occupied):                      # an empty slot can never pass: lo 255 > hi 0
            for a in range(3):
                assert (int(rec["lo"][a]) >> (8 * k)) & 0xFF == 255 and (int(rec["hi"][a]) >> (8 * k)) & 0xFF == 0
    assert sorted(seen_leaves) == sorted(np.nonzero(is_leaf)[0].tolist())
    # every record but the first is some record's slot, once (the frontier checks above found them by their BVH2 node)
    used = [int(r) for rec in wide for r in rec["ref"] if int(r) != EMPTY and not int(r) & LEAF]
    assert sorted(used) == list(range(1, len(wide)))
    return wide


def collapse_cost(nodes, roots):
    return sum(area(nodes, int(n)) for n in roots)


def best_collapse_cost(nodes):
    """The cheapest collapse by exhaustive search: every frontier of at most four nodes below every possible record root."""
    from functools import lru_cache
    is_leaf = (nodes["num_primitives_axis"] >> 16) != 0
    kids = lambda n: (n + 1, int(nodes["offset"][n]))

    def frontiers(n):                                                 # all frontiers of n's subtree with 2..4 nodes
        found, todo = set(), [tuple(kids(n))]
        while todo:
            f = todo.pop()
            if f in found:
                continue
            found.add(f)
            if len(f) < 4:
                for i, c in enumerate(f):
                    if not is_leaf[c]:
                        todo.append(f[:i] + tuple(kids(c)) + f[i + 1:])
        return found

    @lru_cache(maxsize=None)
    def best(n):
        return area(nodes, n) + min(sum(best(c) for c in f if not is_leaf[c]) for f in frontiers(n))
    return best(0)


def test_the_exchange_network_serves_every_shape_a_record_can_fold():
    """Pure combinatorics behind `arrange` in build_wide_bvh: for every binary tree with 2..4 leaves and every assignment of
    split axes to its interior nodes there is a placement of the leaves on the four slot positions such that the four
    conditional exchanges (0,1), (2,3), (0,2), (1,3) produce the reference's visit order for all eight direction octants."""
    import itertools

    def trees(leaves):                                                # every binary tree over the leaf sequence, interior = (id, l, r)
        if len(leaves) == 1:
            yield leaves[0], 0
            return
        for cut in range(1, len(leaves)):
            for l, nl in trees(leaves[:cut]):
                for r, nr in trees(leaves[cut:]):
                    yield ("I", l, r), nl + nr + 1

    def label(t, counter):
        if not isinstance(t, tuple):
            return t
        me = counter[0]; counter[0] += 1
        return (me, label(t[1], counter), label(t[2], counter))

    reach = set()
    for bits in range(16):
        pos = [0, 1, 2, 3]
        for b, (i, j) in enumerate(NETWORK):
            if (bits >> b) & 1:
                pos[i], pos[j] = pos[j], pos[i]
        reach.add(tuple(pos))
    shapes = 0
    for n in (2, 3, 4):
        for t, interior in trees(list(range(n))):
            t = label(t, [0])
            for axes in itertools.product(range(3), repeat=interior):
                def visit(c, octant):
                    if not isinstance(c, tuple):
                        return [c]
                    a, b = (c[2], c[1]) if (octant >> axes[c[0]]) & 1 else (c[1], c[2])
                    return visit(a, octant) + visit(b, octant)
                wanted = {tuple(visit(t, o)) for o in range(8)}
                ok = False
                for place in itertools.permutations(range(4), n):    # place[leaf] = slot position
                    leaf_at = {p: leaf for leaf, p in enumerate(place)}
                    got = {tuple(leaf_at[p] for p in pos if p in leaf_at) for pos in reach}
                    if wanted <= got:
                        ok = True
                        break
                assert ok, (t, axes)
                shapes += 1
    assert shapes == 3 + 2 * 9 + 5 * 27


@pytest.mark.parametrize("seed", range(8))
def test_the_sah_collapse_is_optimal_on_small_trees(seed):
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(5, 120))
    P = (rng.normal(size=(n, 1, 3)) * rng.uniform(0.2, 3.0, size=(1, 1, 3)) + rng.normal(size=(n, 3, 3)) * 0.05).astype(np.float32)
    N = np.tile(np.array([0, 0, 1], np.float32), (n, 3, 1))
    tris = S.to_triangles([(P, N, np.zeros((n, 3, 2), np.float32), 0)])
    mats = np.array([S.make_material(kd=(0.7, 0.7, 0.7))], dtype=T.packed_material)
    nodes, _ = bvh_of(tris, mats)
    if (int(nodes["num_primitives_axis"][0]) >> 16) != 0:
        return
    import sys
    sys.setrecursionlimit(10000)
    _, _, sah_roots = wide_of(nodes, 1, with_roots=True)
    _, _, two_roots = wide_of(nodes, 2, with_roots=True)
    want = best_collapse_cost(nodes)
    got = collapse_cost(nodes, sah_roots)
    assert abs(got - want) <= 1e-9 * want, (got, want)
    assert got <= collapse_cost(nodes, two_roots) * (1 + 1e-12)          # (fewer visits expected, not necessarily fewer records)


def test_wide_tree_of_the_cornell_box_and_a_dense_mesh():
    import os
    s = host.Scene(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets", "CornellBox.obj"))
    s.build_bvh()
    check(s.arrays()["nodes"].copy())
    check(s.arrays()["nodes"].copy(), collapse=2)
    tris, mats = S.cornell_blob(20_000, 2_000)
    nodes, _ = bvh_of(tris, mats)
    n_interior = int(((nodes["num_primitives_axis"] >> 16) == 0).sum())
    wide = check(nodes, collapse=2)
    assert 0.4 * n_interior < len(wide) < 0.75 * n_interior       # two BVH2 levels per record
    sah = check(nodes)
    assert n_interior / 3 <= len(sah) < len(wide)                 # at most three interior nodes folded per record
    assert wide_of(nodes)[0].tobytes() == sah.tobytes()           # (records are made by several host threads: same bytes every time)


@pytest.mark.parametrize("seed", range(6))
def test_wide_tree_of_random_soups_with_extreme_coordinates(seed):
    """Slivers, coincident triangles, huge offsets (coarse fp32 grid) and tiny extents."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 400))
    scale = float(10.0 ** rng.integers(-6, 7))
    offset = rng.normal(size=3) * float(10.0 ** rng.integers(-3, 8))
    P = (rng.normal(size=(n, 1, 3)) * scale + rng.normal(size=(n, 3, 3)) * scale * float(10.0 ** rng.integers(-5, 1)) + offset)
    P = P.astype(np.float32)
    if seed % 2:
        P[: n // 3] = P[0]                                   # coincident triangles
    N = np.tile(np.array([0, 0, 1], np.float32), (n, 3, 1))
    tris = S.to_triangles([(P, N, np.zeros((n, 3, 2), np.float32), 0)])
    mats = np.array([S.make_material(kd=(0.7, 0.7, 0.7))], dtype=T.packed_material)
    nodes, _ = bvh_of(tris, mats)
    check(nodes)
    check(nodes, collapse=2)


def test_trees_that_do_not_qualify_are_refused():
    tris, mats = S.cornell_blob(2_000, 500)
    nodes, _ = bvh_of(tris, mats)
    bad = nodes.copy()
    interior = np.nonzero((bad["num_primitives_axis"] >> 16) == 0)[0]
    bad["bounds_max"]["x"][interior[3] + 1] += 1000.0          # a child sticking out of its parent
    with pytest.raises(capi.RtError, match="does not qualify"):
        wide_of(bad)
    bad = nodes.copy()
    bad["bounds_min"]["y"][5] = np.nan
    with pytest.raises(capi.RtError, match="does not qualify"):
        wide_of(bad)
    # a left-deep chain deeper than the kernel's stack bound is refused, a shallow one qualifies
    def chain(n):
        # interior k at index k (first child k + 1), second child = a leaf stored behind the chain
        c = np.zeros(2 * n + 1, T.bvh_node)
        for k in range(2 * n + 1):
            leaf = k >= n
            c["num_primitives_axis"][k] = (1 << 16) if leaf else 0
            c["offset"][k] = (k - n) if leaf else (2 * n - k)
            for ax in "xyz":
                c["bounds_min"][ax][k] = -0.5 if leaf else -1.0
                c["bounds_max"][ax][k] = 0.5 if leaf else 1.0
        return c
    with pytest.raises(capi.RtError, match="does not qualify"):
        wide_of(chain(80), collapse=2)                         # 40 records deep, two levels each
    with pytest.raises(capi.RtError, match="does not qualify"):
        wide_of(chain(120))                                    # 40 records deep, three levels each
    check(chain(80))
    check(chain(20))
    check(chain(20), collapse=2)


# ---- the one-fma slab distances of k_trace_w4 (loop C), reproduced exactly on the host ------------------------------
from fractions import Fraction


def _rn32(x):
    """Correctly rounded (nearest-even) binary32 of an exact rational, denormals included."""
    if x == 0:
        return np.float32(0.0)
    s = -1 if x < 0 else 1
    x = abs(Fraction(x))
    e = x.numerator.bit_length() - x.denominator.bit_length()         # 2^(e-1) < x < 2^(e+1)
    if Fraction(2) ** e > x:
        e -= 1                                                       # now 2^e <= x < 2^(e+1)
    q = max(e - 23, -149)                                            # weight of the last mantissa bit
    n = x / Fraction(2) ** q
    f = n.numerator // n.denominator
    r = n - f
    if r > Fraction(1, 2) or (r == Fraction(1, 2) and (f & 1)):
        f += 1
    return np.float32(s * float(f) * 2.0 ** q)


def _fma32(a, b, c):
    return _rn32(Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c)))


def _slab_distances_are_conservative(nodes, rays, max_nodes, rng):
    """For every sampled (ray, wide node, slot, axis): the kernel's near distance is <= and its far distance >= what the
    reference's expression fl(fl(plane - org) * inv) (trace_bvh.cl:85-97) gives on the dequantised plane.  The kernel's
    arithmetic (trace_kernels.h, loop C: a = cell * inv, b = fl(fl(origin - org) * inv), m = fma(255, |a|, |b|) + 2^-100,
    near / far offsets fma(-+2^-20, m, b), distance fma(q, a, offset)) is reproduced bit for bit: binary32, every fma
    rounded once (exact rational arithmetic, then one rounding)."""
    f32 = np.float32
    wide, _ = wide_of(nodes)
    if len(wide) == 0:
        return 0
    cells = np.stack([np.ldexp(f32(1.0), (((wide["meta"] >> (8 * a)) & 0xFF).astype(np.int32) - 127)) for a in range(3)], 1).astype(f32)
    pick = rng.permutation(len(wide))[:max_nodes]
    checked = 0
    for org, d in rays:
        org, d = org.astype(f32), d.astype(f32)
        with np.errstate(divide="ignore"):
            inv = (f32(1.0) / d).astype(f32)
        if not (np.isfinite(inv).all() and (np.abs(inv) < f32(2.0) ** 96).all() and (np.abs(org) < f32(2.0) ** 29).all()):
            continue                                                  # RT_SIGN_SLOW / far origin: the BVH2 kernel's rays
        for w in pick:
            rec = wide[w]
            for a in range(3):
                cell, o = cells[w, a], f32(rec["origin"][a])
                A = f32(cell * inv[a])
                assert float(A) == float(cell) * float(inv[a]) or abs(float(A)) < 2.0 ** -120          # exact (power of two)
                b = f32(f32(o - org[a]) * inv[a])
                m = f32(_fma32(f32(255.0), abs(A), abs(b)) + f32(2.0) ** -100)
                bn, bf = _fma32(f32(-(2.0 ** -20)), m, b), _fma32(f32(2.0 ** -20), m, b)
                neg = bool(inv[a] < 0)
                for s in range(4):
                    if int(rec["ref"][s]) == EMPTY:
                        continue
                    qlo, qhi = (int(rec["lo"][a]) >> (8 * s)) & 0xFF, (int(rec["hi"][a]) >> (8 * s)) & 0xFF
                    qn, qf = (qhi, qlo) if neg else (qlo, qhi)
                    pn, pf = f32(f32(qn) * cell + o), f32(f32(qf) * cell + o)                        # exact dequantisation
                    En, Ef = f32(f32(pn - org[a]) * inv[a]), f32(f32(pf - org[a]) * inv[a])
                    Fn, Ff = _fma32(f32(qn), A, bn), _fma32(f32(qf), A, bf)
                    assert Fn <= En and Ff >= Ef, (w, a, s, float(Fn), float(En), float(Ff), float(Ef))
                    checked += 1
    return checked


def _rays_for(nodes, rng, n):
    bmin = np.array([nodes["bounds_min"][c][0] for c in "xyz"], np.float64)
    bmax = np.array([nodes["bounds_max"][c][0] for c in "xyz"], np.float64)
    ext = np.maximum(bmax - bmin, 1e-30)
    rays = []
    for i in range(n):
        org = bmin + rng.uniform(-1.5, 2.5, 3) * ext if i % 3 else bmin + rng.uniform(0, 1, 3) * ext
        d = rng.normal(size=3)
        if i % 4 == 1:
            d[rng.integers(3)] *= 1e-7                                   # nearly parallel to a slab: huge 1/dir
        if i % 7 == 2:
            d[rng.integers(3)] = 1e-30                                   # 1/dir = 1e30: beyond 2^96, skipped like the kernel does
        d /= np.linalg.norm(d)
        rays.append((org, d))
    return rays


def test_one_fma_slab_distances_are_conservative():
    rng = np.random.default_rng(7)
    tris, mats = S.cornell_blob(6_000, 800)
    nodes, _ = bvh_of(tris, mats)
    assert _slab_distances_are_conservative(nodes, _rays_for(nodes, rng, 12), 50, rng) > 3000


@pytest.mark.parametrize("seed", range(4))
def test_one_fma_slab_distances_on_soups_with_extreme_coordinates(seed):
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(8, 200))
    scale = float(10.0 ** rng.integers(-5, 6))
    offset = rng.normal(size=3) * float(10.0 ** rng.integers(-3, 7))
    P = (rng.normal(size=(n, 1, 3)) * scale + rng.normal(size=(n, 3, 3)) * scale * float(10.0 ** rng.integers(-4, 1)) + offset).astype(np.float32)
    N = np.tile(np.array([0, 0, 1], np.float32), (n, 3, 1))
    tris = S.to_triangles([(P, N, np.zeros((n, 3, 2), np.float32), 0)])
    mats = np.array([S.make_material(kd=(0.7, 0.7, 0.7))], dtype=T.packed_material)
    nodes, _ = bvh_of(tris, mats)
    _slab_distances_are_conservative(nodes, _rays_for(nodes, rng, 8), 30, rng)
