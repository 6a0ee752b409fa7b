"""The 4-wide quantized tree of k_trace_w4 (build_wide_bvh in rt_hip.hip), checked on the host:
the invariants the exactness argument of DESIGN.md rests on.

  * every slot box, dequantised EXACTLY in binary32 the way the kernel does it
    (fma(q, cell, origin)), contains the reference node's box;
  * origin + q * cell is exactly representable for q = 0..255 (no rounding in the dequantisation);
  * slots 0,1 / 2,3 are the children of BVH2 child 0 / 1 in the reference's order, the three
    split axes are the reference's, and the leaves reachable from the wide tree are exactly the
    reference's leaves, each once;
  * trees that do not qualify (bounds not nested / not finite) are refused."""
import ctypes as C
import numpy as np
import pytest
from raytracing_amd import capi, host, scenes as S, types as T

LEAF, EMPTY = 0x80000000, 0xFFFFFFFF
WIDE = np.dtype([("origin", "<f4", 3), ("meta", "<u4"), ("lo", "<u4", 3), ("hi", "<u4", 3), ("ref", "<u4", 4), ("order", "<u4"), ("pad", "<u4")])
assert WIDE.itemsize == 64


def wide_of(nodes):
    lib = capi.load()
    n, entry = C.c_uint32(), C.c_uint32()
    nodes = np.ascontiguousarray(nodes)
    rc = lib.rt_debug_wide_bvh(nodes.ctypes.data, len(nodes), None, 0, C.byref(n), C.byref(entry))
    if rc != 0:
        raise capi.RtError(lib.rt_last_error(None).decode())
    out = np.zeros(n.value, WIDE)
    assert lib.rt_debug_wide_bvh(nodes.ctypes.data, len(nodes), out.ctypes.data, len(out), C.byref(n), C.byref(entry)) == 0
    return out, entry.value


def bvh_of(tris, mats):
    s = host.Scene(arrays=dict(triangles=tris, materials=mats))
    s.build_bvh()
    a = s.arrays()
    return a["nodes"].copy(), a["triangles"].copy()


def check(nodes):
    wide, entry = wide_of(nodes)
    is_leaf = (nodes["num_primitives_axis"] >> 16) != 0
    if is_leaf[0]:
        assert len(wide) == 0 and entry == (LEAF | int(nodes["offset"][0]))
        return wide
    assert entry == 0
    bmin = np.stack([nodes["bounds_min"][c] for c in "xyz"], 1)
    bmax = np.stack([nodes["bounds_max"][c] for c in "xyz"], 1)
    # walk the wide tree together with the BVH2: wide node w <-> BVH2 interior node n
    seen_leaves, seen_wide = [], set()
    todo = [(0, 0)]
    f32 = np.float32
    while todo:
        w, n = todo.pop()
        assert w not in seen_wide
        seen_wide.add(w)
        rec = wide[w]
        meta = int(rec["meta"])
        cell = [f32(2.0) ** f32(((meta >> (8 * a)) & 0xFF) - 127) for a in range(3)]
        axes = meta >> 24
        c = [n + 1, int(nodes["offset"][n])]
        assert (axes & 3) == (int(nodes["num_primitives_axis"][n]) & 0xFFFF)
        slots = []
        for i in range(2):
            if is_leaf[c[i]]:
                slots += [c[i], None]
            else:
                slots += [c[i] + 1, int(nodes["offset"][c[i]])]
                assert ((axes >> (2 + 2 * i)) & 3) == (int(nodes["num_primitives_axis"][c[i]]) & 0xFFFF)
        # the order table: per direction-sign octant, swap the halves / inside half 0 / inside half 1 (trace_bvh.cl:181-190)
        ax = [axes & 3, (axes >> 2) & 3, (axes >> 4) & 3]
        for o in range(8):
            want = ((o >> ax[0]) & 1) | ((((o >> ax[1]) & 1) if slots[1] is not None else 0) << 1) | \
                   ((((o >> ax[2]) & 1) if slots[3] is not None else 0) << 2)
            assert (int(rec["order"]) >> (3 * o)) & 7 == want
        for k, child in enumerate(slots):
            ref = int(rec["ref"][k])
            if child is None:
                assert ref == EMPTY
                continue
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
                assert ref == (LEAF | int(nodes["offset"][child]))
                seen_leaves.append(child)
            else:
                assert ref < len(wide)
                todo.append((ref, child))
    assert sorted(seen_leaves) == sorted(np.nonzero(is_leaf)[0].tolist())
    assert len(seen_wide) == len(wide)
    return wide


def test_wide_tree_of_the_cornell_box_and_a_dense_mesh():
    import os
    s = host.Scene(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets", "CornellBox.obj"))
    s.build_bvh()
    check(s.arrays()["nodes"].copy())
    tris, mats = S.cornell_blob(20_000, 2_000)
    nodes, _ = bvh_of(tris, mats)
    wide = check(nodes)
    n_interior = int(((nodes["num_primitives_axis"] >> 16) == 0).sum())
    assert 0.4 * n_interior < len(wide) < 0.75 * n_interior       # two BVH2 levels per record


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
        wide_of(chain(80))
    check(chain(20))


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
