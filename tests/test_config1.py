"""BASELINE.json configs[0] at its stated parameters -- the one config whose asset the repository holds: assets/CornellBox.obj
as shipped, 256 x 256, 1 spp, max_bounces 4, the reference's default camera (camera_controller.cpp:30-41,77-80), the
main.cpp:58 directional light, the CGSkies environment map, sampler kRandom.  SURVEY.md section 8d "Config 1" and Appendix A
publish what the reference's own Scene + Bvh + unmodified kernels produce for sample index 0 (over glibc libm):

    32 triangles, 35 BVH nodes; 265 979 closest-hit + 147 128 shadow rays;
    active rays per bounce 65 536 / 65 536 / 51 957 / 44 567 / 38 383, shadow rays 48 811 / 30 720 / 25 639 / 22 474 / 19 484;
    mean radiance (0.161177, 0.116383, 0.068102).

CPU part: both builds of the reference under oracle/_ref (libm builtins, and the project's rt_detmath.h builtins) and the C
restatement reproduce that table.  GPU part (through the C++ host layer: Scene::Load of the OBJ, Bvh::BuildCPU, the C-ABI): the
HIP path's radiance is bit-identical to the reference kernels' and its queue counters are the table."""
import os
import numpy as np
import pytest
from tests import _oracle, _ref
from raytracing_amd import host, types as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "assets", "CornellBox.obj")
LIGHT = ((-0.6, -1.5, 3.5), (15.0, 10.0, 5.0))
ACTIVE = [65536, 65536, 51957, 44567, 38383]
SHADOW = [48811, 30720, 25639, 22474, 19484]
TOTALS = (265979, 147128)
MEAN = (0.161177, 0.116383, 0.068102)

needs_ref = pytest.mark.skipif(not _ref.available(), reason="oracle/_ref not built (needs /root/reference at build time)")


def reference(libm):
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        sc = _ref.load_scene("assets/CornellBox.obj", dir_lights=[LIGHT], libm=libm)
    finally:
        os.chdir(cwd)
    r = _ref.RefIntegrator(256, 256, sc, libm=libm)
    r.set_camera(T.default_camera(256, 256))
    r.set_max_bounces(4)
    r.integrate(1)
    return sc, r


@needs_ref
@pytest.mark.parametrize("libm", [True, False])
def test_the_reference_builds_reproduce_the_survey_table(libm):
    if libm and not _ref.available(libm=True):
        pytest.skip("libref_libm.so not built")
    sc, r = reference(libm)
    assert len(sc["triangles"]) == 32 and len(sc["nodes"]) == 35
    assert r.ray_totals() == TOTALS
    a, s = r.last_counts(5)
    assert a.tolist() == ACTIVE and s.tolist() == SHADOW
    mean = r.radiance()[..., :3].astype(np.float64).mean((0, 1))
    assert np.allclose(mean, MEAN, rtol=0, atol=5e-7)


@needs_ref
def test_the_c_restatement_reproduces_it_too():
    sc, r = reference(False)
    orc = _oracle.Oracle(256, 256, sc)
    orc.set_camera(T.default_camera(256, 256))
    orc.set_max_bounces(4)
    orc.integrate(1)
    assert orc.ray_totals() == TOTALS
    assert np.array_equal(orc.radiance()[..., :3], r.radiance()[..., :3])


@pytest.mark.gpu
@needs_ref
def test_config_1_on_the_gpu_radiance_and_counts():
    sc, r = reference(False)
    scene = host.Scene(OBJ)                                   # the C++ loader, like `rt_render --scene assets/CornellBox.obj`
    scene.add_directional_light(*LIGHT)
    render = host.Render(256, 256, scene)
    render.set_camera(host.default_camera(256, 256))
    render.set_max_bounces(4)
    render.render_frame()                                     # ONE Integrator::Integrate(): 1 spp
    got = render.radiance()
    assert np.array_equal(got[..., :3], r.radiance()[..., :3]), "radiance differs from the reference's kernels"
    st = render.stats()
    assert (int(st.closest_rays), int(st.shadow_rays)) == TOTALS
    assert [int(x) for x in st.last_active[:5]] == ACTIVE and [int(x) for x in st.last_shadow[:5]] == SHADOW
    mean = got[..., :3].astype(np.float64).mean((0, 1))
    assert np.allclose(mean, MEAN, rtol=0, atol=5e-7)
    # ... and against the libm build (the survey's own probe): within the north star's tolerance, counts equal
    if _ref.available(libm=True):
        _, rl = reference(True)
        want = rl.radiance()[..., :3].astype(np.float64)
        rel = np.linalg.norm(got[..., :3].astype(np.float64) - want) / np.linalg.norm(want)
        assert rel < 1e-4, rel
