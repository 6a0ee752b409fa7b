"""-m gpu: every BASELINE config's stand-in AT ITS OWN SIZE -- frame, triangle count and bounce limit as `bench.py --config N` runs it -- one sample per pixel
through the library's defaults (rt_scene_upload's own choices: device folds, the shadow rays' tree, the adaptation beside the frame) against the reference's own
kernels (oracle/_ref, RefIntegrator: src/kernels/cl/*.cl compiled for the host) on the same scene, camera and sample: bit for bit on the full frame, and the
same ray counts.  VERDICT r05, weak 1 (b): the full-size comparison of configs 2, 3 and 5 was `bench.py`'s (`parity`, builder-run); config 4's is in the
driver's own bench line.  One spp keeps the host side at seconds (config 5, 4K / 16 bounces / 10 M triangles: ~100 M rays on the host's cores)."""
import argparse
import os
import sys
import numpy as np
import pytest
from tests import _ref
from raytracing_amd import capi, host, scenes as S, types as T

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.mark.skipif(not _ref.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("config", [2, 3, 5])
def test_one_sample_of_the_full_frame_equals_the_reference_kernels(config):
    import bench
    cfg = bench.CONFIGS[config]
    args = argparse.Namespace(scene=None, config=config, blob_tris=871_200, ball_tris=20_000)
    scene, n_tris = bench.build_scene(args, host, S)
    scene.build_bvh()
    scene.finalize()
    arrays = scene.arrays()
    w, h, bounces = cfg["width"], cfg["height"], cfg["bounces"]
    assert n_tris >= {2: 800_000, 3: 800_000, 5: 9_000_000}[config]
    cam = T.default_camera(w, h)
    ctx = capi.Context(0)
    ctx.upload_scene(arrays)
    fr = capi.Frame(ctx, w, h)
    fr.set_camera(cam)
    fr.set_max_bounces(bounces)
    fr.integrate(1)
    got = fr.radiance()[..., :3]
    st = fr.stats()
    fr.close()
    ctx.close()
    ri = _ref.RefIntegrator(w, h, arrays, threads=min(32, os.cpu_count() or 1))
    ri.set_camera(cam)
    ri.set_max_bounces(bounces)
    ri.integrate(1)
    want = ri.radiance()[..., :3]
    diff = ~((got == want) | (np.isnan(got) & np.isnan(want))).all(-1)
    assert not diff.any(), "%d of %d pixels differ, first at %s" % (diff.sum(), diff.size, np.argwhere(diff)[:3].tolist())
    assert (st.closest_rays, st.shadow_rays) == ri.ray_totals()
