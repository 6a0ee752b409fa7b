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
