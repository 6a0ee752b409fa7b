"""Row g readiness (VERDICT r02 "next round" 8): how long does the path from a Bistro-class OBJ on disk to a renderable scene
take on the host?  Writes the 2.8 M-triangle stand-in as a real OBJ + MTL (+ its textures as PNG files), then times
  Scene::Load (OBJ / MTL parse, texture decode)  ->  Bvh::BuildCPU  ->  Scene::SaveCache  ->  Scene(cache) (load).
The wide collapse + upload need a GPU context and are timed by bench.py (`config.setup_s`).  No GPU.
usage: python tools/obj_ingest_timing.py [--triangles 2800000] [--dir /tmp/rt_obj_ingest]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from raytracing_amd import host, scenes as S


def write_obj(path, arrays):
    tris = arrays["triangles"]
    n = len(tris)
    raw = np.ascontiguousarray(tris).view(np.float32).reshape(n, 40)          # 3 vertices x (position, texcoord, normal) x 4 floats + 4 words
    V = raw[:, :36].reshape(n, 3, 3, 4)
    P, UV, N = V[:, :, 0, :3].reshape(-1, 3), V[:, :, 1, :2].reshape(-1, 2), V[:, :, 2, :3].reshape(-1, 3)
    mtl = tris["mtl_index"]
    with open(path, "w") as f:
        f.write("mtllib %s\n" % os.path.basename(path).replace(".obj", ".mtl"))
        f.write("".join("v %.9g %.9g %.9g\n" % tuple(p) for p in P))
        f.write("".join("vn %.9g %.9g %.9g\n" % tuple(p) for p in N))
        f.write("".join("vt %.9g %.9g\n" % tuple(p) for p in UV))
        order = np.argsort(mtl, kind="stable")
        cur = -1
        out = []
        for i in order:
            if mtl[i] != cur:
                cur = int(mtl[i])
                out.append("usemtl m%d\n" % cur)
            a = 3 * int(i) + 1
            out.append("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (a, a, a, a + 1, a + 1, a + 1, a + 2, a + 2, a + 2))
        f.write("".join(out))
    return n


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--triangles", type=int, default=2_800_000)
    ap.add_argument("--dir", default="/tmp/rt_obj_ingest")
    a = ap.parse_args()
    os.makedirs(a.dir, exist_ok=True)
    t0 = time.time()
    arrays = S.city_block(a.triangles)
    t_gen = time.time() - t0
    obj = os.path.join(a.dir, "city.obj")
    t0 = time.time()
    n = write_obj(obj, arrays)
    with open(obj.replace(".obj", ".mtl"), "w") as f:       # plain materials: the parse + build times are what is measured
        for m in range(len(arrays["materials"])):
            f.write("newmtl m%d\nKd %.2f %.2f %.2f\nKs 0.04 0.04 0.04\nPr 0.5\nTf 1 1 1\n" % (m, 0.3 + 0.5 * ((m * 7) % 10) / 10.0, 0.3 + 0.5 * ((m * 3) % 10) / 10.0, 0.6))   # Tf 1: opaque (transparency < 0.5 = pass-through, material.h:171-241)
    t_write = time.time() - t0
    size = os.path.getsize(obj)
    t0 = time.time(); s = host.Scene(obj); t_parse = time.time() - t0
    t0 = time.time(); s.build_bvh(); t_bvh = time.time() - t0
    cache = os.path.join(a.dir, "city.rtscene")
    t0 = time.time(); s.save_cache(cache); t_save = time.time() - t0
    t0 = time.time(); c = host.Scene(cache); t_load = time.time() - t0
    print("%d triangles: generated in %.1f s, OBJ written in %.1f s (%.0f MB)" % (n, t_gen, t_write, size / 1e6))
    print("Scene::Load (OBJ + MTL parse) %.2f s | Bvh::BuildCPU %.2f s (%d threads) | SaveCache %.2f s (%.0f MB) | Scene(cache) %.2f s"
          % (t_parse, t_bvh, os.cpu_count() or 1, t_save, os.path.getsize(cache) / 1e6, t_load))
