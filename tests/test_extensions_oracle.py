"""CPU: the restatement of the opt-in emissive-triangle next-event estimation (oracle.c: Light_SampleWithEmissive;
the HIP kernels are compared with it bit for bit in tests/test_gpu_extensions.py) against a closed form.  A Lambertian
floor under a small square emitter, camera rays only (max_bounces = 0): every floor pixel must show
Le * brdf * cos_s * cos_l * A / d^2 with the reference's own brdf for direct light (material.h:132-169: diffuse scaled
by 1 - Schlick(f0, h.o), no glossy lobe at roughness 0) -- which pins the triangle selection probability, the area
sampling and the geometry term in one number (the emitter faces the floor: like every triangle of the reference, whose
traversal culls back faces, it is visible and emits on the side its normal points to only)."""
import numpy as np
from tests import _oracle
from raytracing_amd import host, scenes as S, types as T


def test_emissive_nee_estimator_matches_the_closed_form():
    kd, Le, H, a = 0.5, 100.0, 3.0, 0.05
    mats = [S.make_material(kd=(kd, kd, kd)), S.make_material(kd=(0, 0, 0), ke=(Le, Le, Le))]
    meshes = [S.quad((-4, -4, 0), (4, -4, 0), (4, 4, 0), (-4, 4, 0)) + (0,),
              S.quad((-a, -a, H), (-a, a, H), (a, a, H), (a, -a, H)) + (1,)]
    s = host.Scene(arrays=dict(triangles=S.to_triangles(meshes), materials=np.array(mats, dtype=T.packed_material),
                               textures=np.zeros(0, T.texture), texture_data=np.zeros(0, np.uint32)))
    s.build_bvh(); s.set_env_image(np.zeros((8, 16, 4), np.float32)); s.finalize()
    sc = s.arrays()
    assert len(sc["emissive"]) == 2 and len(sc["lights"]) == 0
    sc["flags"] = 1                                            # RT_SCENE_EMISSIVE_NEE
    w, h, spp = 40, 30, 256
    cam = T.default_camera(w, h)
    for field, vec in (("position", (0.0, 0.0, 2.0)), ("front", (0.0, 0.0, -1.0)), ("up", (0.0, 1.0, 0.0))):
        for k, x in zip("xyz", vec):
            cam[field][k] = np.float32(x)
    cam["fov"] = np.float32(1.0)
    orc = _oracle.Oracle(w, h, sc)
    orc.set_camera(cam); orc.set_max_bounces(0); orc.integrate(spp)
    img = orc.radiance()[..., 0] / spp
    # the same pixels in closed form (pinhole camera of raygeneration.cl:100-118)
    t = np.tan(0.5 * float(cam["fov"]))
    ys, xs = np.mgrid[0:h, 0:w]
    u, v = ((xs + 0.5) / w * 2 - 1) * t * (w / h), ((ys + 0.5) / h * 2 - 1) * t
    right = np.cross([0, 0, -1.0], [0, 1.0, 0])
    d = np.array([0, 0, -1.0]) + right * u[..., None] + np.array([0, 1.0, 0]) * v[..., None]
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    P = np.array([0, 0, 2.0]) + d * (-2.0 / d[..., 2])[..., None]
    to_l = np.array([0, 0, H]) - P
    d2 = (to_l ** 2).sum(-1)
    wo = to_l / np.sqrt(d2)[..., None]
    hv = -d + wo
    hv /= np.linalg.norm(hv, axis=-1, keepdims=True)
    f0 = ((1.5 - 1.0) / (1.5 + 1.0)) ** 2                     # IorToF0(1, 1.5), bxdf.h:57-61
    F = f0 + (1 - f0) * (1 - (hv * wo).sum(-1)) ** 5
    kd_eff = float(sc["materials"][0]["diffuse_albedo"] & 0xFF) / 255.0      # what UnpackRGBTex returns
    want = Le * (1 - F) * kd_eff / np.pi * wo[..., 2] * wo[..., 2] * (2 * a) ** 2 / d2
    ratio = img / want
    assert abs(img.sum() / want.sum() - 1.0) < 0.01, img.sum() / want.sum()
    assert 0.97 < ratio.min() and ratio.max() < 1.03, (ratio.min(), ratio.max())
    # without the extension the same render shows no direct light at all: the floor is black at max_bounces = 0
    off = dict(sc); off["flags"] = 0
    orc0 = _oracle.Oracle(w, h, off)
    orc0.set_camera(cam); orc0.set_max_bounces(0); orc0.integrate(4)
    assert float(orc0.radiance()[..., :3].max()) == 0.0
