"""-m gpu: seeded random scenes / cameras / lights / options, HIP path vs the C oracle,
bit for bit (radiance, ray counters).  Every case is small enough for the single-threaded
oracle; together they walk combinations the hand-written scenes do not: random triangle
soups with sliver and coincident triangles, all material parameter ranges incl. textures
on every slot, point lights inside geometry, axis-aligned cameras and lights (zero
direction components -> inf / NaN in the slab test), depth of field, 0..9 bounces,
furnace, both samplers, several samples in flight and kernel variants."""
import os
import numpy as np
import pytest
from tests import _oracle
from raytracing_amd import capi, host, scenes as S, types as T

pytestmark = pytest.mark.gpu


def random_scene(rng, env):
    n_tex = int(rng.integers(0, 4))
    textures = np.zeros(n_tex, dtype=T.texture)
    data = []
    start = 0
    for i in range(n_tex):
        w, h = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        textures[i] = (start, w, h, 0)
        data.append(rng.integers(0, 2 ** 32, w * h, dtype=np.uint64).astype(np.uint32))
        start += w * h
    texture_data = np.concatenate(data).astype(np.uint32) if data else np.zeros(0, np.uint32)

    def tex():
        return int(rng.integers(0, n_tex)) if n_tex and rng.random() < 0.3 else S.INVALID_TEX
    mats = []
    for _ in range(int(rng.integers(1, 9))):
        mats.append(S.make_material(
            kd=tuple(rng.uniform(0, 1, 3)), ks=tuple(rng.uniform(0, 1, 3)) if rng.random() < 0.7 else (0, 0, 0),
            ke=tuple(rng.uniform(0, 20, 3)) if rng.random() < 0.2 else (0, 0, 0),
            roughness=float(rng.choice([0.0, 0.005, 0.01, rng.uniform(0, 1)])), metalness=float(rng.choice([0.0, 1.0, rng.uniform(0, 1)])),
            ior=float(rng.choice([1.0, 1.5, rng.uniform(0.5, 4)])), transparency=float(rng.choice([1.0, 1.0, 0.0, rng.uniform(0, 1)])),
            kd_tex=tex(), ks_tex=tex(), r_tex=tex(), m_tex=tex(), e_tex=tex(), t_tex=tex()))
    materials = np.array(mats, dtype=T.packed_material)

    meshes = []
    if rng.random() < 0.7:                                   # a floor / some boxes: axis-aligned planes
        meshes.append(S.quad((-2, -1, 0), (2, -1, 0), (2, 3, 0), (-2, 3, 0), uv_scale=float(rng.uniform(0.5, 4))) + (0,))
        for _ in range(int(rng.integers(0, 4))):
            c = rng.uniform([-1, 0, 0], [1, 2, 1]); e = rng.uniform(0.05, 0.5, 3)
            meshes.append(S.box(c - e, c + e) + (int(rng.integers(0, len(mats))),))
    for _ in range(int(rng.integers(0, 3))):
        meshes.append(S.uv_sphere(tuple(rng.uniform([-1, 0, 0.2], [1, 2, 1.5])), float(rng.uniform(0.1, 0.6)),
                                  int(rng.integers(3, 14)), int(rng.integers(3, 20)), bump=float(rng.uniform(0, 0.4)),
                                  seed=int(rng.integers(1, 1000))) + (int(rng.integers(0, len(mats))),))
    n = int(rng.integers(1, 400))                            # soup: random, sliver, coincident triangles
    P = (rng.uniform(-1, 1, (n, 1, 3)) * [1.5, 1.5, 1.0] + [0, 1, 0.8] + rng.normal(0, 0.25, (n, 3, 3))).astype(np.float32)
    k = n // 5
    P[:k, 2] = P[:k, 1] + (P[:k, 1] - P[:k, 0]) * 1e-3      # slivers (det ~ 0)
    if n > 8:
        P[k:k + 4] = P[k]                                    # coincident: equal centroids, equal t
    Nn = rng.normal(0, 1, (n, 3, 3)).astype(np.float32)
    Nn /= np.maximum(np.linalg.norm(Nn, axis=2, keepdims=True), 1e-6)
    U = rng.uniform(-3, 3, (n, 3, 2)).astype(np.float32)
    tris = np.concatenate([S.to_triangles(meshes) if meshes else np.zeros(0, T.triangle),
                           S.to_triangles([(P, Nn, U, 0)])])
    tris["mtl_index"][-n:] = rng.integers(0, len(mats), n)

    s = host.Scene(arrays=dict(triangles=tris, materials=materials, textures=textures, texture_data=texture_data))
    for _ in range(int(rng.integers(0, 3))):
        d = rng.normal(0, 1, 3)
        if rng.random() < 0.4:                               # axis-aligned light: zero direction components
            d = np.eye(3)[rng.integers(0, 3)] * rng.choice([-1.0, 1.0])
        s.add_directional_light(tuple(d), tuple(rng.uniform(0, 12, 3)))
    for _ in range(int(rng.integers(0, 3))):
        s.add_point_light(tuple(rng.uniform([-1.5, -0.5, 0], [1.5, 2.5, 2])), tuple(rng.uniform(0, 6, 3)))
    s.build_bvh()
    s.set_env_image(env)
    s.finalize()
    return s.arrays()


def _set3(cam, field, v):
    for k, x in zip("xyz", v):
        cam[field][k] = np.float32(x)


def random_camera(rng, w, h):
    cam = T.default_camera(w, h)
    if rng.random() < 0.5:
        _set3(cam, "position", rng.uniform([-1, -2, 0.2], [1, 0, 2]))
    mode = rng.integers(0, 3)
    if mode == 1:                                            # exactly along an axis
        _set3(cam, "front", np.eye(3)[rng.integers(0, 2)] * rng.choice([-1.0, 1.0]))
        _set3(cam, "up", (0.0, 0.0, 1.0))
    elif mode == 2:
        f = rng.normal(0, 1, 3); f[1] = abs(f[1]) + 0.5; f /= np.linalg.norm(f)
        r = np.cross(f, [0, 0, 1.0]); r /= np.linalg.norm(r)
        _set3(cam, "front", f)
        _set3(cam, "up", np.cross(r, f))
    if rng.random() < 0.3:
        cam["aperture"] = float(rng.uniform(0, 0.1)); cam["focus_distance"] = float(rng.uniform(0.5, 4))
    cam["fov"] = np.float32(rng.uniform(0.3, 2.2))
    return cam


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    c.upload_blue_noise_tables(*S.blue_noise_tables())
    yield c
    c.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("RT_FUZZ_FIRST", "0")), int(os.environ.get("RT_FUZZ_SEEDS", "24"))))   # RT_FUZZ_SEEDS=2000 for a campaign (RT_FUZZ_FIRST: seeds below it are skipped)
def test_random_scene_matches_oracle_bit_for_bit(ctx, env_map, seed):
    rng = np.random.default_rng(1000 + seed)
    sc = random_scene(rng, env_map)
    # the opt-in extensions on some seeds (keyed by the seed, so that the random stream of the other choices stays as it was)
    if seed % 5 == 4 and len(sc["emissive"]):
        sc["flags"] = capi.SCENE_EMISSIVE_NEE
    if seed % 7 == 3:
        m = sc["materials"]
        idx = np.stack([m["diffuse_albedo"] >> 24, m["specular_albedo"] >> 24, (m["roughness_metalness"] >> 8) & 0xFF,
                        m["roughness_metalness"] >> 24, (m["ior_emission_idx_transparency"] >> 8) & 0xFF,
                        m["ior_emission_idx_transparency"] >> 24], axis=1).astype(np.uint32)
        sc["material_texture_indices"] = np.where(idx == 0xFF, 0xFFFF, idx).astype(np.uint16)
    w, h = int(rng.integers(8, 112)), int(rng.integers(8, 80))
    cam = random_camera(rng, w, h)
    bounces = int(rng.integers(0, 10))
    spp = int(rng.integers(1, 9))
    furnace = bool(rng.random() < 0.2)
    blue = bool(rng.random() < 0.3)
    ctx.set_wide_bvh(2 if seed % 4 == 1 else 1)                       # the wide tree's collapse: SAH-optimal (default) / two BVH2 levels per record
    # the shadow rays' tree (round 4; keyed by the seed: the random stream of the other choices stays as it was): measured choice
    # (default) / the backend's own tree forced / own with the surface-area metric / shared with the closest-hit rays -- same bits
    ctx.set_shadow_tree((1, 2, 3, 2, 0)[seed % 5])
    # RT_CTX_OPT_ADAPTIVE_FOLD on every sixth seed (7 = on, wait for the new fold, small trees too): the first integrate() below then traces a
    # probe frame, both trees are folded again for its rays and the records replaced before the frame's own rays start -- same bits
    ctx.set_adaptive_fold((7, 15, 31)[(seed // 6) % 3] if seed % 6 == 5 else (0, capi.ADAPTIVE_FOLD_DEFAULT)[seed % 2])   # 15: + the shadow tree rotated first, 31: + slots occluder first
    try:
        ctx.upload_scene(sc)
    finally:
        ctx.set_wide_bvh(1)
        ctx.set_shadow_tree(1)
        ctx.set_adaptive_fold(capi.ADAPTIVE_FOLD_DEFAULT)
    fr = capi.Frame(ctx, w, h)
    fr.set_camera(cam); fr.set_max_bounces(bounces)
    applied = []                                                       # (the frame's options, for the tiles of the seeds that render in tiles)
    def opt(o, v):
        applied.append((o, v))
        fr.set_option(o, v)
    opt(capi.OPT_WHITE_FURNACE, int(furnace))
    opt(capi.OPT_SAMPLER, int(blue))
    opt(capi.OPT_SAMPLES_IN_FLIGHT, int(rng.integers(0, 5)))
    variant = int(rng.choice([0, 8, 5, 9, 8, 9, 10, 10, 10, 11]))
    opt(capi.OPT_TRACE_VARIANT, int(os.environ.get("RT_FUZZ_VARIANT", variant)))   # a campaign on one kernel
    opt(capi.OPT_TRACE_TUNE, int(rng.choice([0, (2 << 24) | (1 << 23), 24 | (4 << 8), 56 | (32 << 8) | (7 << 24), 64 | (1 << 8) | (1 << 23)])))
    opt(capi.OPT_SHADE_PARTITION, int(seed & 3))            # bit 0: hits first, bit 1: outputs grouped by octant
    if seed % 3 == 0:                                                  # the compact radiance log, with a pool small enough to run dry now and then
        opt(capi.OPT_COMPACT_LOG, 1)
        opt(capi.OPT_DEBUG_LOG_POOL_DIV, 8 if seed % 2 else 64)
        if seed % 9 == 0:
            opt(capi.OPT_SAMPLES_IN_FLIGHT, 8)
    # k_trace_w4's loop D (round 4): the instance that has it for every launch / none, and when it takes over a wave's last lanes
    opt(capi.OPT_TRACE_TAIL_PATHS, (4000000000, 0, 50000000)[seed % 3])
    opt(capi.OPT_TRACE_TAIL_LANES, (40, 1, 64, 16, 0)[(seed // 3) % 5])
    opt(capi.OPT_CHUNK_REFILL, 0 if seed % 4 == 2 else 1)    # chunk mode: round 3's form / lanes refilled from the wave's own chunks
    opt(capi.OPT_SMALL_LAUNCH_PATHS, int(rng.choice([3000000, 0, 4000000000, 700])))   # chunk mode below this many rays per launch: default, never, always, for the last bounces
    # k_frame (round 5): every fifth seed renders its samples through the stage API with RT_OPT_FRAME_KERNEL (1: every block resident, 2 / 3: that
    # many chunks per wave) -- one launch per sample in which each wave carries its own pixels through all the bounces; where the frame is not
    # eligible (compact log, a forced kernel variant) the same calls take the stage kernels
    # RT_OPT_SAMPLES_AHEAD (round 6): every seventh seed renders at least six samples through the stage API with the next samples traced ahead in batches
    # of 2 / 3 (one stream, or one per bank) and replayed one per call -- the sum must be the oracle's all the same
    ahead = seed % 7 == 3 and seed % 5 != 2
    if ahead:
        spp = max(spp, 6)
        fr.set_option(capi.OPT_SAMPLES_AHEAD, (2, 3, 256 + 2, 64)[(seed // 7) % 4])
        for _ in range(spp):
            fr.generate_rays()
            for bounce in range(bounces + 1):
                fr.intersect(bounce); fr.shade(bounce); fr.intersect_shadow(bounce)
            fr.advance_sample()
    elif seed % 5 == 2:
        fr.set_option(capi.OPT_FRAME_KERNEL, (1, 2, 3)[(seed // 5) % 3])
        for _ in range(spp):
            fr.generate_rays()
            for bounce in range(bounces + 1):
                fr.intersect(bounce); fr.shade(bounce); fr.intersect_shadow(bounce)
            fr.advance_sample()
    else:
        # RT_OPT_PATH_STATE_LIMIT_MB on every eleventh of these seeds (round 6): 1 MB of path state -- a tile of more than 4096 pixels is then rendered in chunks, one
        # after the other, and batches of 8 samples in flight take the compact log on their own (the stage calls refuse a limit below one sample of the whole tile)
        if seed % 11 == 7:
            opt(capi.OPT_PATH_STATE_LIMIT_MB, 1)
            opt(capi.OPT_PIPELINES, 1 + (seed // 11) % 3)               # ... the chunks on 1 - 3 pipes (streams of their own; more than one: the full log layout)
        if seed % 11 in (3, 7):
            opt(capi.OPT_OVERLAP_SHADOW, (seed // 11) % 2)             # the shadow trace of bounce b beside the next bounce (the default) or in line
        # every thirteenth of these seeds renders the frame as 2 - 4 TILES (interleaved bands of 1 / 2 / 4 / 8 rows: rt_frame_desc, the multi-GPU path's
        # decomposition) with the same options, one frame per tile on this device; the assembled rows and the summed ray counts must be the whole frame's
        if seed % 13 == 5:
            tiles, band = 2 + (seed // 13) % 3, (1, 2, 4, 8)[(seed // 39) % 4]
            tiled = np.zeros((h, w, 3), np.float32)
            tiled_rays = [0, 0]
            for rank in range(tiles):
                t = capi.Frame(ctx, w, h, tile_rank=rank, tile_count=tiles, band_height=band)
                t.set_camera(cam); t.set_max_bounces(bounces)
                for o, v in applied:
                    t.set_option(o, v)
                t.integrate(spp)
                rows = t.global_rows()
                if len(rows):
                    tiled[rows] = t.radiance()[..., :3]
                ts = t.stats()
                tiled_rays[0] += ts.closest_rays; tiled_rays[1] += ts.shadow_rays
                t.close()
        fr.integrate(spp)
    orc = _oracle.Oracle(w, h, sc, furnace=furnace)
    orc.set_camera(cam); orc.set_max_bounces(bounces)
    orc.set_blue_noise(blue, S.blue_noise_tables())
    orc.integrate(spp)
    got, want = fr.radiance()[..., :3], orc.radiance()[..., :3]
    assert np.array_equal(got, want, equal_nan=True), (seed, np.argwhere(~np.isclose(got, want, equal_nan=True))[:4])
    st = fr.stats()
    if ahead:
        assert st.samples_from_banks == spp - 3 and st.samples == spp, (seed, st.samples_from_banks, spp)      # (the banks' ray totals run ahead: rt_stats.samples_ahead)
    else:
        assert (st.closest_rays, st.shadow_rays) == orc.ray_totals()
    if not ahead and seed % 5 != 2 and seed % 13 == 5:
        assert np.array_equal(tiled, want, equal_nan=True), (seed, "tiles", np.argwhere(~np.isclose(tiled, want, equal_nan=True))[:4])
        assert tuple(tiled_rays) == orc.ray_totals(), (seed, "tiles")
