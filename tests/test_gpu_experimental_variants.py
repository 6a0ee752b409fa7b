"""-m gpu, OPT-IN (RT_TEST_EXPERIMENTAL=1): kernel instances that are in the library but have not been through a full GPU
validation yet -- they are never selected automatically and these tests are how they get validated before they may be.

  RT_OPT_TRACE_VARIANT 15   k_trace_w4<.., DIRECT>: the first slot that passes its box test is visited next instead of being
                            pushed to the LDS stack and popped right back.  Same node sequence by construction; the CPU
                            restatement of both forms equals the reference loop (tests/test_wide_traversal_oracle.py).
                            Round 2 ran out of GPU budget before it could be run on hardware.
A campaign on it: RT_TEST_EXPERIMENTAL=1 RT_FUZZ_VARIANT=15 RT_FUZZ_SEEDS=2000 pytest tests/test_gpu_fuzz.py -m gpu -n 32;
the whole suite on it: RT_TRACE_AUTO_WIDE_VARIANT=15 pytest tests -m gpu; speed: python bench.py --trace-variant 15."""
import os
import numpy as np
import pytest
from tests import _oracle
from raytracing_amd import capi, types as T

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("RT_TEST_EXPERIMENTAL") != "1", reason="opt-in: RT_TEST_EXPERIMENTAL=1")]


@pytest.mark.parametrize("slots", [1, 3, 8])
@pytest.mark.parametrize("tune", [0, 64 | (64 << 8), 1 | (1 << 8)])
def test_direct_visit_instance_equals_the_oracle(golden_scenes, slots, tune):
    w, h, b, spp = 96, 72, 6, 7
    for name in ("coverage", "cornell"):
        sc = golden_scenes[name]
        cam = T.default_camera(w, h)
        ctx = capi.Context(0)
        ctx.upload_scene(sc)
        fr = capi.Frame(ctx, w, h)
        fr.set_camera(cam); fr.set_max_bounces(b)
        fr.set_option(capi.OPT_SAMPLES_IN_FLIGHT, slots)
        fr.set_option(capi.OPT_TRACE_VARIANT, 15)
        fr.set_option(capi.OPT_TRACE_TUNE, tune)
        fr.integrate(spp)
        orc = _oracle.Oracle(w, h, sc)
        orc.set_camera(cam); orc.set_max_bounces(b); orc.integrate(spp)
        assert np.array_equal(fr.radiance()[..., :3], orc.radiance()[..., :3], equal_nan=True), (name, slots, tune)
        st = fr.stats()
        assert (st.closest_rays, st.shadow_rays) == orc.ray_totals()
        fr.close(); ctx.close()
