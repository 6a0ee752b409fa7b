"""One seed of tests/test_gpu_fuzz.py rendered with one of its choices changed at a time: which choice a failing seed needs.
    python tools/fuzz_seed_probe.py 5652"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import _oracle
from tests.test_gpu_fuzz import random_scene, random_camera
from raytracing_amd import capi, scenes as S

seed = int(sys.argv[1])
from raytracing_amd import host
ENV = host.load_hdr(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets", "ibl", "CGSkies_0036_free.hdr"))


def run(over):
    rng = np.random.default_rng(1000 + seed)
    sc = random_scene(rng, ENV)
    if seed % 5 == 4 and len(sc["emissive"]):
        sc["flags"] = capi.SCENE_EMISSIVE_NEE
    w, h = int(rng.integers(8, 112)), int(rng.integers(8, 80))
    cam = random_camera(rng, w, h)
    bounces = int(rng.integers(0, 10)); spp = int(rng.integers(1, 9))
    furnace = bool(rng.random() < 0.2); blue = bool(rng.random() < 0.3)
    ctx = capi.Context(0)
    ctx.upload_blue_noise_tables(*S.blue_noise_tables())
    ctx.set_wide_bvh(over.get("wide", 2 if seed % 4 == 1 else 1))
    ctx.set_shadow_tree(over.get("shadow_tree", (1, 2, 3, 2, 0)[seed % 5]))
    ctx.set_adaptive_fold(over.get("adaptive", (7, 15, 31)[(seed // 6) % 3] if seed % 6 == 5 else (0, capi.ADAPTIVE_FOLD_DEFAULT)[seed % 2]))
    for opt, val in over.get("ctx", ()):
        assert capi.load().rt_ctx_set_option(ctx.handle, opt, val) == 0
    ctx.upload_scene(sc)
    fr = capi.Frame(ctx, w, h)
    fr.set_camera(cam); fr.set_max_bounces(bounces)
    fr.set_option(capi.OPT_WHITE_FURNACE, int(furnace)); fr.set_option(capi.OPT_SAMPLER, int(blue))
    fr.set_option(capi.OPT_SAMPLES_IN_FLIGHT, int(rng.integers(0, 5)))
    variant = int(rng.choice([0, 8, 5, 9, 8, 9, 10, 10, 10, 11]))
    fr.set_option(capi.OPT_TRACE_VARIANT, over.get("variant", variant))
    tune = int(rng.choice([0, (2 << 24) | (1 << 23), 24 | (4 << 8), 56 | (32 << 8) | (7 << 24), 64 | (1 << 8) | (1 << 23)]))
    fr.set_option(capi.OPT_TRACE_TUNE, over.get("tune", tune))
    fr.set_option(capi.OPT_SHADE_PARTITION, over.get("partition", int(seed & 3)))
    compact = over.get("compact", seed % 3 == 0)
    if compact:
        fr.set_option(capi.OPT_COMPACT_LOG, 1)
        fr.set_option(capi.OPT_DEBUG_LOG_POOL_DIV, over.get("pool_div", 8 if seed % 2 else 64))
        if seed % 9 == 0:
            fr.set_option(capi.OPT_SAMPLES_IN_FLIGHT, over.get("in_flight", 8))
    fr.set_option(capi.OPT_TRACE_TAIL_PATHS, over.get("tail_paths", (4000000000, 0, 50000000)[seed % 3]))
    fr.set_option(capi.OPT_TRACE_TAIL_LANES, over.get("tail_lanes", (40, 1, 64, 16, 0)[(seed // 3) % 5]))
    fr.set_option(capi.OPT_CHUNK_REFILL, over.get("refill", 0 if seed % 4 == 2 else 1))
    small = int(rng.choice([3000000, 0, 4000000000, 700]))
    fr.set_option(capi.OPT_SMALL_LAUNCH_PATHS, over.get("small", small))
    stage = over.get("stage", seed % 5 == 2)
    if stage:
        fr.set_option(capi.OPT_FRAME_KERNEL, over.get("frame_kernel", (1, 2, 3)[(seed // 5) % 3]))
        for _ in range(spp):
            fr.generate_rays()
            for bounce in range(bounces + 1):
                fr.intersect(bounce); fr.shade(bounce); fr.intersect_shadow(bounce)
            fr.advance_sample()
    else:
        fr.integrate(spp)
    orc = _oracle.Oracle(w, h, sc, furnace=furnace)
    orc.set_camera(cam); orc.set_max_bounces(bounces); orc.set_blue_noise(blue, S.blue_noise_tables()); orc.integrate(spp)
    got, want = fr.radiance()[..., :3], orc.radiance()[..., :3]
    bad = ~((got == want) | (np.isnan(got) & np.isnan(want))).all(-1)
    st = fr.stats()
    info = dict(w=w, h=h, bounces=bounces, spp=spp, furnace=furnace, blue=blue, variant=variant, tune=tune, small=small, tris=len(sc["triangles"]), nodes=len(sc["nodes"]),
                fallbacks=getattr(st, "log_fallbacks", None), inline=getattr(st, "log_inline_entries", None), in_flight=st.samples_in_flight)
    fr.close(); ctx.close()
    return int(bad.sum()), np.argwhere(bad)[:4].tolist(), info


n, where, info = run({})
print("seed", seed, "as the test renders it:", n, "pixels differ", where, info)
for name, over in [("integrate() instead of the stage API", dict(stage=False)), ("frame kernel 0", dict(frame_kernel=0)), ("no compact log", dict(compact=False)),
                   ("pool div 8", dict(pool_div=8)), ("in flight 1", dict(in_flight=1)), ("shadow tree 1", dict(shadow_tree=1)), ("shadow tree 0", dict(shadow_tree=0)), ("variant 0", dict(variant=0)), ("variant 8", dict(variant=8)),
                   ("tune 0", dict(tune=0)), ("partition 0", dict(partition=0)), ("tail lanes 0", dict(tail_lanes=0)), ("tail paths 0", dict(tail_paths=0)), ("refill 1", dict(refill=1)),
                   ("small 3000000", dict(small=3000000)), ("small 0", dict(small=0)), ("device fold off", dict(ctx=((7, 0),))), ("tree builder host", dict(ctx=((9, 0),)))]:
    try:
        n, where, info2 = run(over)
        print("%-40s %d pixels differ %s fallbacks %s" % (name, n, where, info2["fallbacks"]))
    except Exception as e:
        print("%-40s ERROR %r" % (name, e))
