import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Native pieces are built once per session (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g
    g.build()
    os.chdir(ROOT)


GOLDEN = os.path.join(ROOT, "tests", "golden")
SCENE_KEYS = ("triangles", "nodes", "materials", "textures", "texture_data", "lights", "emissive")


def load_golden_scene(name, env):
    from raytracing_amd import types as T
    z = np.load(os.path.join(GOLDEN, name + "_scene.npz"))
    dts = dict(triangles=T.triangle, nodes=T.bvh_node, materials=T.packed_material, textures=T.texture,
               texture_data=np.uint32, lights=T.light, emissive=np.uint32)
    sc = {k: z[k].astype(dts[k], copy=False) for k in SCENE_KEYS}
    sc["env"] = env
    return sc


@pytest.fixture(scope="session")
def env_map():
    """The reference's environment map decoded by the PRODUCT loader (rt::LoadHDR);
    its bytes are pinned against the reference loader's hash in test_host_layer.py."""
    from raytracing_amd import host
    return host.load_hdr(os.path.join(ROOT, "assets", "ibl", "CGSkies_0036_free.hdr"))


@pytest.fixture(scope="session")
def golden_scenes(env_map):
    return {"cornell": load_golden_scene("cornell", env_map), "coverage": load_golden_scene("coverage", env_map)}


@pytest.fixture(scope="session")
def golden_radiance():
    return np.load(os.path.join(GOLDEN, "radiance.npz"))


GOLDEN_CASES = [
    # name, scene, width, height, bounces, spp, furnace
    ("cornell_64_b4_s2", "cornell", 64, 64, 4, 2, False),
    ("cornell_96x64_b2_s1", "cornell", 96, 64, 2, 1, False),
    ("coverage_64_b6_s2", "coverage", 64, 64, 6, 2, False),
    ("coverage_64_b6_s2_furnace", "coverage", 64, 64, 6, 2, True),
    ("coverage_80x48_b5_s2_dof", "coverage", 80, 48, 5, 2, False),
    ("coverage_48_b0_s1", "coverage", 48, 48, 0, 1, False),
]
