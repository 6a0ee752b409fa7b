"""The CPU-only analysis tools stay runnable: the path-regeneration model and the wave-schedule replay (both feed numbers
DESIGN.md quotes).  Small inputs; no GPU."""
import importlib.util
import os
import numpy as np
from tests import _oracle
from tests.test_wide_bvh import wide_of
from raytracing_amd import host, scenes as S, types as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_regeneration_model_orders_the_budgets():
    m = _load("regeneration_model")
    s16, ms16 = m.chunked(16.0)
    s133, ms133 = m.chunked(133.0)
    assert s16 == 15 and s133 == 127 and ms133 < ms16                      # more samples in flight: less time per sample
    r = m.regeneration(16.0, 16, np.random.default_rng(0), pixels=512, rounds=150)
    assert r is not None and 0.3 < r["fill"] <= 1.0 and r["ms_per_spp"] > 0


def test_wave_schedule_replay_counts_every_step(env_map):
    m = _load("wave_schedule_model")
    scene = host.Scene(arrays=S.city_block(20_000))
    scene.add_directional_light((-0.6, -1.5, 3.5), (15.0, 10.0, 5.0))
    scene.build_bvh(); scene.set_env_image(env_map); scene.finalize()
    arrays = scene.arrays()
    wide, entry = wide_of(arrays["nodes"])
    w, h = 64, 36
    orc = _oracle.Oracle(w, h, arrays)
    orc.set_camera(T.default_camera(w, h)); orc.set_max_bounces(2)
    orc.stage("reset"); orc.stage("generate_rays")
    rays = orc.buffer("rays0", T.ray, w * h)
    ev, ln = orc.wide_trace_events(wide, entry, rays, False)
    steps = int(np.minimum(ln, ev.shape[1]).sum())
    for node_q, leaf_q in ((32, 8), (1, 1), (64, 64)):
        passes, lanes = m.replay(ev, ln, node_q, leaf_q)
        assert lanes["B"] + lanes["C"] == steps                            # every step of every ray is executed exactly once
        assert passes["B"] + passes["C"] >= steps / 64.0
    p32, _ = m.replay(ev, ln, 32, 8)
    p1, _ = m.replay(ev, ln, 1, 1)
    assert p32["B"] + p32["C"] < p1["B"] + p1["C"]                          # the thresholds are what fills the passes


def test_code_object_identity_and_the_stale_flag_of_the_roofline(tmp_path):
    """profiles/r0X_trace_counters.json names the code object its counters were collected from; bench.py's `roofline`
    says `stale` when the library now running is another one ("cannot tell when the file is stale")."""
    import json
    import bench
    from raytracing_amd import codeobj
    digest = codeobj.code_object_sha256()
    assert digest is not None and len(digest) == 64 and digest == codeobj.code_object_sha256()
    assert codeobj.code_object_sha256(str(tmp_path / "missing.so")) is None
    k = dict(avg_launch_ms=14.7, hbm_bytes_per_launch=8.0e9, hbm_read_bytes_per_launch=5.0e9, hbm_write_bytes_per_launch=3.0e9, hbm_GBs=8.0e9 / 14.7e-3 / 1e9,
             valu_busy=0.86, salu_busy=0.5, l1_ta_busy=0.79, hbm_frac=0.068,
             cycles_per_launch=3.45e7, per_launch=dict(l1_accesses=7.6e9, l2_hit_rate=0.86), launches_profiled=27)
    live = dict(avg_launch_ms=15.0, rays_per_launch=1.0e8, mrays_per_s=6666.7, kernel_ms_per_spp=dict(trace_closest=1.05, trace_shadow=0.5, shade=0.4, raygen=0.02))
    per_ray = dict(closest_nodes=77.7, closest_tris=2.9, closest_steps=24.07, closest_wide_visits=20.9)
    iso = dict(avg_launch_ms=14.7, rays_per_launch=1.0e8, kernel_ms_per_spp=dict(trace_closest=1.03, trace_shadow=0.36, shade=0.37, raygen=0.02))
    old = bench.COUNTERS_FILE
    try:
        # ... or when the counters were collected on another FOLD of the trees than the run walks (round 4: the same code object visits fewer
        # records on a fold adapted to the frame's rays; a file without the field is of the surface-area fold)
        for recorded, fold, adaptive, want_stale in ((digest, "adapted to the frame's rays", 3, False), ("0" * 64, "adapted to the frame's rays", 3, True),
                                                     (digest, None, 0, False), (digest, None, 3, True), (digest, "adapted to the frame's rays", 0, True)):
            args = type("A", (), dict(config=4, adaptive_fold=adaptive))()
            bench.COUNTERS_FILE = str(tmp_path / "counters.json")
            doc = {"config_4": {"closest": k}, "_code_object_sha256": recorded, "_how": "test"}
            if fold is not None:
                doc["_fold"] = fold
            json.dump(doc, open(bench.COUNTERS_FILE, "w"))
            r = bench.roofline_object(args, 1, live, per_ray, iso)
            assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
            assert abs(r["achieved"] - 8.0e9 / 14.7e-3 / 1e9) < 0.1 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-4     # the stated formula
            assert r["traffic"] == 8.0e9 and r["stale"] is want_stale and r["counters"]["stale"] is want_stale
            assert r["units"]["valu_busy"] == 0.86 and r["algorithmic"]["traffic_over_algorithmic"] < 0.1
            assert "latency_ceiling" not in r                 # (a "ceiling" the kernel exceeded: gone from the line)
            c = r["ceilings"]
            assert abs(c["grays"]["valu_issue"] - c["achieved_grays"] / 0.86) < 2e-3 and abs(c["grays"]["l1_texture_address"] - c["achieved_grays"] / 0.79) < 2e-3
            assert set(c["grays"]) == {"valu_issue", "l1_texture_address", "hbm"} and "frac_of_ceiling" not in c
            assert c["binding"] == min(c["grays"], key=c["grays"].get) and abs(c["busiest_unit_fraction"] - c["achieved_grays"] / min(c["grays"].values())) < 1e-3
            # every kernel of the path against HBM and against the calibration kernels' rates (round 6)
            h = r["hbm_all_kernels"]
            assert h["kernels"]["closest"]["read_GB"] == 5.0 and h["measured_ceilings"]["random_64B_records_line_traffic_TBs"] > h["measured_ceilings"]["random_64B_records_useful_TBs"]
        bench.COUNTERS_FILE = str(tmp_path / "none.json")
        r = bench.roofline_object(args, 1, live, per_ray, iso)
        assert r["achieved"] is None and "error" in r["counters"]                 # says so instead of silently changing `bound`
    finally:
        bench.COUNTERS_FILE = old
