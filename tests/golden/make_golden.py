"""Regenerates the golden fixtures in this directory from the REFERENCE build
(oracle/_ref/libref.so = the reference's unmodified .cl kernels + its own
Scene/Bvh/LoadHDR compiled for x86-64 by oracle/Makefile).  Needs
/root/reference (run in the build container):

    python tests/golden/make_golden.py

The reference ships no tests or golden vectors of its own;
these files are the pins.  Outputs:
  cornell_scene.npz    assets/CornellBox.obj through the reference Scene + Bvh
  coverage_scene.npz   raytracing_amd.scenes.coverage_scene() through the reference Bvh
  radiance.npz         running-sum radiance + ray counters of the reference kernels
  host.npz             hashes of the reference LoadHDR output, camera record
"""
import hashlib
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
from tests import _ref                      # noqa: E402
from raytracing_amd import types as T, scenes as S   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
LIGHT = [((-0.6, -1.5, 3.5), (15.0, 10.0, 5.0))]   # main.cpp:58


def clean(a):
    """Zero the padding members so the fixture bytes are deterministic."""
    a = a.copy()

    def walk(arr, dt):
        for name in dt.names:
            sub = dt.fields[name][0]
            if name in ("w", "padding", "pad"):
                arr[name] = 0
            elif sub.names:
                walk(arr[name], sub)
    if a.dtype.names:
        walk(a, a.dtype)
    return a


CASES = [
    # name, scene key, width, height, bounces, spp, furnace, aperture, focus
    ("cornell_64_b4_s2", "cornell", 64, 64, 4, 2, False, 0.0, 10.0),
    ("cornell_96x64_b2_s1", "cornell", 96, 64, 2, 1, False, 0.0, 10.0),
    ("coverage_64_b6_s2", "coverage", 64, 64, 6, 2, False, 0.0, 10.0),
    ("coverage_64_b6_s2_furnace", "coverage", 64, 64, 6, 2, True, 0.0, 10.0),
    ("coverage_80x48_b5_s2_dof", "coverage", 80, 48, 5, 2, False, 0.05, 1.5),
    ("coverage_48_b0_s1", "coverage", 48, 48, 0, 1, False, 0.0, 10.0),
]


def main():
    env = _ref.load_hdr("assets/ibl/CGSkies_0036_free.hdr")
    cornell = _ref.load_scene("assets/CornellBox.obj", dir_lights=LIGHT)
    cov = S.coverage_scene()
    cov["triangles"], cov["nodes"] = _ref.bvh_build(cov["triangles"])
    cov["emissive"] = np.zeros(0, np.uint32)
    scenes = {"cornell": dict(cornell), "coverage": cov}
    for s in scenes.values():
        s["env"] = env
    keys = ("triangles", "nodes", "materials", "textures", "texture_data", "lights", "emissive")
    np.savez_compressed(os.path.join(HERE, "cornell_scene.npz"), **{k: clean(cornell[k]) for k in keys})
    np.savez_compressed(os.path.join(HERE, "coverage_scene.npz"), **{k: clean(cov[k]) for k in keys})

    out = {}
    for name, key, w, h, b, spp, furnace, ap, focus in CASES:
        sc = scenes[key]
        ri = _ref.RefIntegrator(w, h, sc, furnace=furnace, threads=1)
        cam = T.default_camera(w, h)
        cam["aperture"] = ap
        cam["focus_distance"] = focus
        ri.set_camera(cam)
        ri.set_max_bounces(b)
        ri.integrate(spp)
        out[name + "/radiance"] = ri.radiance()[..., :3].copy()
        out[name + "/resolved"] = ri.resolve()[..., :3].copy()
        c, s = ri.ray_totals()
        a, sh = ri.last_counts(b + 1)
        out[name + "/totals"] = np.array([c, s], np.uint64)
        out[name + "/last_active"] = a
        out[name + "/last_shadow"] = sh
        out[name + "/camera"] = cam
        print(name, "mean", out[name + "/radiance"].mean(), "rays", c, s)
    # AOV viewer + temporal denoiser (aov.cl, denoiser.cl, resolve_radiance.cl AOV switch): a 5-frame
    # sequence with a moving camera, resolved image of every frame
    w, h = 64, 48
    ri = _ref.RefIntegrator(w, h, scenes["coverage"], threads=1)
    ri.set_max_bounces(3)
    cam = T.default_camera(w, h)
    for aov in (1, 2, 3, 4):
        ri.set_aov(aov)
        ri.set_camera(cam)
        ri.integrate(1)
        out["aov%d/resolved" % aov] = ri.resolve()[..., :3].copy()
    ri.set_aov(0)
    ri.enable_denoiser(True)
    for f in range(5):
        c = cam.copy()
        c["position"]["x"] = 0.02 * f
        ri.set_camera(c)
        ri.integrate(1)
        out["denoise%d/resolved" % f] = ri.resolve()[..., :3].copy()
        out["denoise%d/radiance" % f] = ri.radiance()[..., :3].copy()
    out["aov_denoise/camera"] = cam
    # blue-noise sampler (hit_surface.cl -D BLUE_NOISE_SAMPLER), frame larger than the 128x128 tile
    bw, bh = 160, 136
    bcam = T.default_camera(bw, bh)
    for furnace in (False, True):
        ri = _ref.RefIntegrator(bw, bh, scenes["coverage"], furnace=furnace, threads=1)
        ri.set_camera(bcam)
        ri.set_max_bounces(9)
        ri.set_blue_noise(True, S.blue_noise_tables())
        ri.integrate(3)
        out["blue_noise/radiance_furnace%d" % furnace] = ri.radiance()[..., :3].copy()
    out["blue_noise/camera"] = bcam
    np.savez_compressed(os.path.join(HERE, "radiance.npz"), **out)

    host = {
        "env_sha256": np.frombuffer(hashlib.sha256(env.tobytes()).digest(), np.uint8),
        "env_shape": np.array(env.shape, np.int64),
        "default_camera_1280x720": T.default_camera(1280, 720),
    }
    np.savez_compressed(os.path.join(HERE, "host.npz"), **host)


if __name__ == "__main__":
    main()
