"""Scratch GPU sanity run: HIP path vs oracle (and vs oracle/_ref when present)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import _ref, _oracle
from raytracing_amd import types as T, scenes as S, capi

def cmp(name, a, b):
    a = a[..., :3]; b = b[..., :3]
    eq = np.array_equal(a, b)
    nd = int((a != b).any(-1).sum())
    rel = float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b.astype(np.float64)), 1e-30))
    print(f"{name}: equal={eq} differing_px={nd} relL2={rel:.3e} mean={a.mean():.6f}/{b.mean():.6f}", flush=True)
    return eq

ctx = capi.Context(0)
print(ctx.device_info(), flush=True)

# device math KATs
rng = np.random.RandomState(1)
orc = _oracle.load()
x = (rng.rand(200000) * 6.2831855).astype(np.float32)
for fn, name, of in ((0, "sin", orc.orc_sinf), (1, "cos", orc.orc_cosf)):
    g = ctx.debug_eval(fn, x)
    c = np.array([of(float(v)) for v in x[:20000]], np.float32)
    print(name, "mismatch", int((g[:20000].view(np.uint32) != c.view(np.uint32)).sum()), flush=True)
u = rng.rand(20000).astype(np.float32)
g = ctx.debug_eval(3, u, np.full_like(u, 2.2)); c = np.array([orc.orc_powf(float(v), 2.2) for v in u], np.float32)
print("pow2.2 mismatch", int((g.view(np.uint32) != c.view(np.uint32)).sum()))
a1 = (rng.rand(20000) * 2 - 1).astype(np.float32); a2 = (rng.rand(20000) * 2 - 1).astype(np.float32)
g = ctx.debug_eval(4, a1, a2); c = np.array([orc.orc_atan2f(float(p), float(q)) for p, q in zip(a1, a2)], np.float32)
print("atan2 mismatch", int((g.view(np.uint32) != c.view(np.uint32)).sum()))
g = ctx.debug_eval(5, a1); c = np.array([orc.orc_acosf(float(p)) for p in a1], np.float32)
print("acos mismatch", int((g.view(np.uint32) != c.view(np.uint32)).sum()))
g = ctx.debug_eval(6, u); print("sqrt mismatch", int((g.view(np.uint32) != np.sqrt(u).view(np.uint32)).sum()))
g = ctx.debug_eval(7, a1, a2); print("div mismatch", int((g.view(np.uint32) != (a1 / a2).view(np.uint32)).sum()))
y = (rng.rand(20000) * 0.999).astype(np.float32)
g = ctx.debug_eval(9, u, y)
c = (1.0 / np.sqrt(1.0 + u.astype(np.float64) / (1.0 - y.astype(np.float64)))).astype(np.float32)
print("ggx fp64 mismatch", int((g.view(np.uint32) != c.view(np.uint32)).sum()), flush=True)

env = _ref.load_hdr("assets/ibl/CGSkies_0036_free.hdr")
sc1 = _ref.load_scene("assets/CornellBox.obj", dir_lights=[((-0.6, -1.5, 3.5), (15., 10., 5.))])
sc2 = S.coverage_scene()
sc2["triangles"], sc2["nodes"] = _ref.bvh_build(sc2["triangles"])
sc2["emissive"] = np.zeros(0, np.uint32); sc2["env"] = env

ok = True
for name, sc, n, b, spp in (("cornell", sc1, 128, 4, 4), ("coverage", sc2, 128, 6, 4)):
    for furnace in (False, True):
        ctx.upload_scene(sc)
        fr = capi.Frame(ctx, n, n)
        cam = T.default_camera(n, n)
        fr.set_camera(cam); fr.set_max_bounces(b); fr.set_option(capi.OPT_WHITE_FURNACE, int(furnace))
        t = time.time(); fr.integrate(spp); g = fr.radiance(); dt = time.time() - t
        o = _oracle.Oracle(n, n, sc, furnace=furnace); o.set_camera(cam); o.set_max_bounces(b); o.integrate(spp)
        st = fr.stats()
        print(name, "furnace", furnace, "gpu rays", st.closest_rays, st.shadow_rays, "oracle", o.ray_totals(), "%.3fs" % dt)
        ok &= cmp(f"{name} furnace={furnace} HIP vs oracle", g, o.radiance())
        fr.close()
# tiling invariance
ctx.upload_scene(sc2)
full = capi.Frame(ctx, 96, 64); cam = T.default_camera(96, 64)
full.set_camera(cam); full.set_max_bounces(5); full.integrate(3); F = full.radiance()
img = np.zeros_like(F)
for r in range(3):
    t = capi.Frame(ctx, 96, 64, tile_rank=r, tile_count=3, band_height=8)
    t.set_camera(cam); t.set_max_bounces(5); t.integrate(3)
    img[t.global_rows()] = t.radiance()
ok &= cmp("tiled(3) vs full", img, F)
print("ALL OK" if ok else "MISMATCH", flush=True)

# quick throughput probe
tris, mats = S.cornell_blob(200000, 20000)
tris, nodes = _ref.bvh_build(tris)
scb = dict(triangles=tris, nodes=nodes, materials=mats, textures=np.zeros(0, T.texture), texture_data=np.zeros(0, np.uint32),
           lights=S.make_lights(directional=[((-0.6, -1.5, 3.5), (15., 10., 5.))]), emissive=np.zeros(0, np.uint32), env=env)
ctx.upload_scene(scb)
fr = capi.Frame(ctx, 1280, 720); fr.set_camera(T.default_camera(1280, 720)); fr.set_max_bounces(8)
fr.integrate(2); ctx.finish()
fr.reset(); t = time.time(); fr.integrate(8); ctx.finish(); dt = time.time() - t
st = fr.stats()
print("blob 1280x720 B=8 8spp: %.3fs  rays %d+%d  %.1f Mrays/s" % (dt, st.closest_rays, st.shadow_rays, (st.closest_rays + st.shadow_rays) / dt / 1e6))
print("per-bounce", list(st.last_active[:9]), list(st.last_shadow[:9]))
