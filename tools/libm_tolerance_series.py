"""How far is the HIP path from the reference's kernels when those run over ANOTHER conformant builtin library?  (VERDICT r03,
What's weak 1.)  The arithmetic contract pins the transcendentals OpenCL leaves implementation-defined to rt_detmath.h -- used by the
HIP kernels, by oracle.c and by the shim under the reference's kernels, which is why those three agree bit for bit.  The one pin that does
NOT share it is oracle/_ref/libref_libm.so: the same unmodified reference kernels over glibc's libm.  A 1-ulp difference in a
sine can flip a hit, and a flipped path moves its pixel by O(1): on the deep-foliage stand-in of config 5 that is rel-L2 8e-4 at
2 spp.  Flips are independent from sample to sample, so the distance should fall like spp^-1/2; this tool MEASURES it:
the same frame, the same sample indices 0 .. n-1, HIP on the GPU against libref_libm.so on the host cores, at a series of sample
counts; fitted slope in log-log; the sample count at which the series crosses the north star's 1e-4.

    python tools/libm_tolerance_series.py                       # 960x540, 16 bounces, config 5's foliage (10 M triangles), 2 / 8 / 32 / 128 spp
    python tools/libm_tolerance_series.py --cpu-only ...        # no GPU: libref.so (rt_detmath.h builtins, == HIP bit for bit) stands in for it
Prints one JSON object."""
import argparse, json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

LIGHT = ((-0.6, -1.5, 3.5), (15.0, 10.0, 5.0))


def rel_l2(a, b):
    fin = np.isfinite(a).all(-1) & np.isfinite(b).all(-1)
    den = np.linalg.norm(b[fin].astype(np.float64).ravel())
    return float(np.linalg.norm((a[fin].astype(np.float64) - b[fin]).ravel()) / den) if den > 0 else 0.0


def fit(spps, vals):
    """least squares of log(rel_l2) = log(c) + slope * log(spp); the spp where it crosses 1e-4"""
    pts = [(math.log(s), math.log(v)) for s, v in zip(spps, vals) if v > 0]
    if len(pts) < 2:
        return None, None, None
    n = len(pts); sx = sum(p[0] for p in pts); sy = sum(p[1] for p in pts)
    sxx = sum(p[0] * p[0] for p in pts); sxy = sum(p[0] * p[1] for p in pts)
    slope = (n * sxy - sx * sy) / (n * sxx - sx * sx)
    logc = (sy - slope * sx) / n
    cross = math.exp((math.log(1e-4) - logc) / slope) if slope < 0 else None
    return slope, math.exp(logc), cross


def series(scene_arrays, width, height, bounces, spps, cam, hip_image_at, threads):
    """hip_image_at(n) -> the HIP (or stand-in) radiance sum / n after samples 0 .. n-1.  Returns the list of points."""
    from tests import _ref
    rl = _ref.RefIntegrator(width, height, scene_arrays, threads=threads, libm=True)
    rl.set_camera(cam); rl.set_max_bounces(bounces)
    out, done = [], 0
    for n in spps:
        t0 = time.time()
        rl.integrate(n - done); done = n
        ref = rl.radiance()[..., :3] / np.float32(n)
        got = hip_image_at(n)[..., :3] / np.float32(n)
        differ = int((~((got == ref) | (np.isnan(got) & np.isnan(ref))).all(-1)).sum())
        # where the distance sits: the share of the squared difference the worst pixel / the worst 16 pixels carry (a path that
        # flips and ends on a mirror chain towards the sun moves ONE pixel by thousands of units: the contributions are heavy-tailed)
        fin = np.isfinite(got).all(-1) & np.isfinite(ref).all(-1)
        d2 = ((got.astype(np.float64) - ref) ** 2).sum(-1)[fin]
        top = np.sort(d2)[::-1]
        tot = float(d2.sum())
        # ... and the same comparison on what the reference DISPLAYS: sum / spp through its Reinhard curve x / (1 + x)
        # (resolve_radiance.cl:78-84), where one pixel can move by at most 1
        tm = lambda x: x / (1.0 + x)
        out.append(dict(spp=n, rel_l2=rel_l2(got, ref), rel_l2_resolved=rel_l2(tm(got.astype(np.float64)), tm(ref.astype(np.float64))),
                        differing_pixels=differ, nan_pixels=(int((~np.isfinite(got).all(-1)).sum()), int((~np.isfinite(ref).all(-1)).sum())),
                        nan_positions_equal=bool(np.array_equal(np.isfinite(got).all(-1), np.isfinite(ref).all(-1))),
                        share_of_worst_pixel=round(float(top[0]) / tot, 4) if tot > 0 else 0.0,
                        share_of_worst_16_pixels=round(float(top[:16].sum()) / tot, 4) if tot > 0 else 0.0,
                        largest_pixel_difference=float(np.sqrt(top[0])) if len(top) else 0.0,
                        # one firefly cannot move a median: |difference| / max(|reference|, 1e-6) per pixel, the middle pixel's
                        median_pixel_rel_err=float(np.median(np.sqrt(d2) / np.maximum(np.linalg.norm(ref.astype(np.float64), axis=-1)[fin], 1e-6))) if fin.any() else None,
                        seconds=round(time.time() - t0, 1)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=960)
    ap.add_argument("--height", type=int, default=540)
    ap.add_argument("--bounces", type=int, default=16)
    ap.add_argument("--triangles", type=int, default=10_000_000)
    ap.add_argument("--scene", default="foliage", choices=("foliage", "city"))
    ap.add_argument("--spp", default="2,8,32,128")
    ap.add_argument("--threads", type=int, default=0, help="host threads of the reference leg (0 = min(64, all))")
    ap.add_argument("--cpu-only", action="store_true", help="libref.so (rt_detmath.h builtins; bit-identical to the HIP path) instead of the GPU")
    a = ap.parse_args()
    from raytracing_amd import host, scenes as S
    from tests import _ref
    if not _ref.available(libm=True):
        print(json.dumps(dict(error="oracle/_ref/libref_libm.so is not built")))
        return 1
    spps = sorted(int(x) for x in a.spp.split(","))
    threads = a.threads or min(64, os.cpu_count() or 1)
    scene = host.Scene(arrays={"foliage": S.dense_foliage, "city": S.city_block}[a.scene](a.triangles))
    scene.add_directional_light(*LIGHT)
    scene.set_env_path(os.path.join(ROOT, "assets", "ibl", "CGSkies_0036_free.hdr"))
    cam = host.default_camera(a.width, a.height)
    if a.cpu_only:
        scene.build_bvh(); scene.finalize()
        arrays = scene.arrays()
        rd = _ref.RefIntegrator(a.width, a.height, arrays, threads=threads)
        rd.set_camera(cam); rd.set_max_bounces(a.bounces)
        state = dict(done=0)
        def hip_image_at(n):
            rd.integrate(n - state["done"]); state["done"] = n
            return rd.radiance()
        who = "oracle/_ref/libref.so (the reference's kernels over rt_detmath.h: bit-identical to the HIP path)"
    else:
        render = host.Render(a.width, a.height, scene)
        render.set_camera(cam); render.set_max_bounces(a.bounces)
        arrays = render.scene_arrays()
        state = dict(done=0)
        def hip_image_at(n):
            render.render_samples(n - state["done"]); state["done"] = n
            return render.radiance()
        who = "the HIP path on the GPU"
    pts = series(arrays, a.width, a.height, a.bounces, spps, cam, hip_image_at, threads)
    slope, c, cross = fit([p["spp"] for p in pts], [p["rel_l2"] for p in pts])
    slope_r, c_r, cross_r = fit([p["spp"] for p in pts], [p["rel_l2_resolved"] for p in pts])
    print(json.dumps(dict(resolved=dict(fitted_slope=slope_r, fitted_rel_l2_at_1_spp=c_r, crosses_1e_4_at_spp=cross_r,
                                        what="the same fit on rel_l2_resolved (the displayed image: Reinhard of sum / spp)"),
                          what="rel-L2 of %s against oracle/_ref/libref_libm.so (the same kernels over glibc libm), same frame, same sample indices" % who,
                          scene="%s stand-in, %d triangles" % (a.scene, len(arrays["triangles"])), width=a.width, height=a.height, bounces=a.bounces,
                          series=pts, fitted_slope=slope, fitted_rel_l2_at_1_spp=c, crosses_1e_4_at_spp=cross, tolerance=1e-4,
                          reading="flipped paths are independent from sample to sample: a slope near -0.5 is what that predicts")))
    return 0


if __name__ == "__main__":
    sys.exit(main())
