"""Procedural stand-in scenes (host-side plumbing, numpy only).

Four of the five BASELINE.json configs name meshes that are not available
offline (CornellBox_Dragon.obj and ShaderBalls.obj are
stripped from the reference checkout, Bistro is a download script).  This
module generates deterministic substitutes with the same material tables
(assets/CornellBox_Dragon.mtl, assets/ShaderBalls.mtl values) and comparable
triangle counts, either as in-memory `rt_triangle`/`rt_packed_material` arrays
or as OBJ/MTL text for the scene loader.

Material packing follows the reference's host packer (src/scene/scene.cpp:53-124:
sRGB->linear pow 2.2, truncation to 8 bit, RGBE emission, ior*25.5).
"""
import math
import numpy as np
from . import types as T

INVALID_TEX = 0xFF


# --------------------------------------------------------------------------
# material packing (scene.cpp:53-124, 150-186)
# --------------------------------------------------------------------------
def _clamp(x, lo, hi):
    return lo if x < lo else (hi if x > hi else x)


def pack_albedo(rgb, tex=INVALID_TEX, gamma=True):
    f32 = np.float32
    out = 0
    for i, c in enumerate(rgb):
        c = f32(c)
        if gamma:
            c = f32(math.pow(float(c), float(f32(2.2))))
        c = f32(_clamp(c, f32(0), f32(1)))
        out |= int(f32(c * f32(255.0))) << (8 * i)
    return out | (tex << 24)


def pack_rgbe(rgb):
    f32 = np.float32
    r, g, b = (max(f32(c), f32(0)) for c in rgb)
    v = max(r, g, b)
    if v < 1e-32:
        return 0
    m, e = math.frexp(float(v))
    scale = f32(f32(f32(m) * f32(256.0)) / f32(v))
    return int(f32(r * scale)) | (int(f32(g * scale)) << 8) | (int(f32(b * scale)) << 16) | ((e + 128) << 24)


def pack_roughness_metalness(roughness, metalness, rtex=INVALID_TEX, mtex=INVALID_TEX):
    f32 = np.float32
    r = int(f32(_clamp(f32(roughness), f32(0), f32(1)) * f32(255.0)))
    m = int(f32(_clamp(f32(metalness), f32(0), f32(1)) * f32(255.0)))
    return r | (rtex << 8) | (m << 16) | (mtex << 24)


def pack_ior_transparency(ior, transparency, etex=INVALID_TEX, ttex=INVALID_TEX):
    f32 = np.float32
    i = int(f32(_clamp(f32(ior), f32(0), f32(10)) * f32(25.5)))
    t = int(f32(_clamp(f32(transparency), f32(0), f32(1)) * f32(255.0)))
    return i | (etex << 8) | (t << 16) | (ttex << 24)


def make_material(kd=(0.7, 0.7, 0.7), ks=(0, 0, 0), ke=(0, 0, 0), roughness=0.0, metalness=0.0, ior=1.5,
                  transparency=1.0, kd_tex=INVALID_TEX, ks_tex=INVALID_TEX, r_tex=INVALID_TEX, m_tex=INVALID_TEX,
                  e_tex=INVALID_TEX, t_tex=INVALID_TEX):
    m = np.zeros((), dtype=T.packed_material)
    m["diffuse_albedo"] = pack_albedo(kd, kd_tex)
    m["specular_albedo"] = pack_albedo(ks, ks_tex)
    m["emission"] = pack_rgbe(ke)
    m["roughness_metalness"] = pack_roughness_metalness(roughness, metalness, r_tex, m_tex)
    m["ior_emission_idx_transparency"] = pack_ior_transparency(ior, transparency, e_tex, t_tex)
    return m


# --------------------------------------------------------------------------
# mesh helpers: every mesh is (positions[n,3,3], normals[n,3,3], uvs[n,3,2])
# --------------------------------------------------------------------------
def _tri_arrays(n):
    return (np.zeros((n, 3, 3), np.float32), np.zeros((n, 3, 3), np.float32), np.zeros((n, 3, 2), np.float32))


def quad(p0, p1, p2, p3, normal=None, uv_scale=1.0):
    """Two triangles (p0,p1,p2), (p2,p3,p0); front face = counter-clockwise."""
    P, N, U = _tri_arrays(2)
    pts = np.array([p0, p1, p2, p3], np.float32)
    if normal is None:
        n = np.cross(pts[1] - pts[0], pts[2] - pts[0])
        normal = n / np.linalg.norm(n)
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32) * uv_scale
    for t, idx in enumerate(((0, 1, 2), (2, 3, 0))):
        for k, i in enumerate(idx):
            P[t, k] = pts[i]
            N[t, k] = normal
            U[t, k] = uv[i]
    return P, N, U


def box(lo, hi):
    lo = np.asarray(lo, np.float32)
    hi = np.asarray(hi, np.float32)
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    faces = [
        ((x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1)),   # +z
        ((x0, y1, z0), (x1, y1, z0), (x1, y0, z0), (x0, y0, z0)),   # -z
        ((x1, y0, z0), (x1, y1, z0), (x1, y1, z1), (x1, y0, z1)),   # +x
        ((x0, y1, z0), (x0, y0, z0), (x0, y0, z1), (x0, y1, z1)),   # -x
        ((x1, y1, z0), (x0, y1, z0), (x0, y1, z1), (x1, y1, z1)),   # +y
        ((x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1)),   # -y
    ]
    parts = [quad(*f) for f in faces]
    return tuple(np.concatenate([p[i] for p in parts]) for i in range(3))


def uv_sphere(center, radius, n_lat, n_lon, bump=0.0, bump_freq=7.0, seed=1234):
    """Tessellated sphere with smooth normals; `bump` displaces the radius by a
    deterministic sum of sines (a cheap high-poly 'blob')."""
    center = np.asarray(center, np.float64)
    th = np.linspace(0.0, np.pi, n_lat + 1)
    ph = np.linspace(0.0, 2.0 * np.pi, n_lon + 1)
    TH, PH = np.meshgrid(th, ph, indexing="ij")
    d = np.stack([np.sin(TH) * np.cos(PH), np.sin(TH) * np.sin(PH), np.cos(TH)], -1)
    rng = np.random.RandomState(seed)
    k = rng.uniform(-1, 1, size=(4, 3)) * bump_freq
    ph0 = rng.uniform(0, 2 * np.pi, size=4)
    r = np.full(TH.shape, radius, np.float64)
    if bump:
        for i in range(4):
            r += bump * radius * 0.25 * np.sin(d @ k[i] + ph0[i])
    pos = center + d * r[..., None]
    # smooth normals by finite differences of the displaced surface
    if bump:
        dth = np.gradient(pos, axis=0)
        dph = np.gradient(pos, axis=1)
        nrm = np.cross(dth, dph)
        ln = np.linalg.norm(nrm, axis=-1, keepdims=True)
        nrm = np.where(ln > 1e-12, nrm / np.maximum(ln, 1e-12), d)
        nrm = np.where((np.sum(nrm * d, -1, keepdims=True) < 0), -nrm, nrm)
    else:
        nrm = d
    uv = np.stack([PH / (2 * np.pi), 1.0 - TH / np.pi], -1)
    i, j = np.meshgrid(np.arange(n_lat), np.arange(n_lon), indexing="ij")
    i = i.ravel()
    j = j.ravel()
    a, b, c, dd = (i, j), (i + 1, j), (i + 1, j + 1), (i, j + 1)

    def gather(arr, tri):
        return np.stack([arr[t[0], t[1]] for t in tri], 1)

    P = np.concatenate([gather(pos, (a, b, c)), gather(pos, (a, c, dd))]).astype(np.float32)
    N = np.concatenate([gather(nrm, (a, b, c)), gather(nrm, (a, c, dd))]).astype(np.float32)
    U = np.concatenate([gather(uv, (a, b, c)), gather(uv, (a, c, dd))]).astype(np.float32)
    # drop the degenerate cap triangles
    e1 = P[:, 1] - P[:, 0]
    e2 = P[:, 2] - P[:, 0]
    area = np.linalg.norm(np.cross(e1, e2), axis=-1)
    keep = area > 1e-12
    return P[keep], N[keep], U[keep]


def to_triangles(meshes):
    """[(P,N,U, material_index)] -> rt_triangle array (reference Triangle, 160 B)."""
    n = sum(len(m[0]) for m in meshes)
    tris = np.zeros(n, dtype=T.triangle)
    o = 0
    for P, N, U, mtl in meshes:
        k = len(P)
        sl = tris[o:o + k]
        for vi, vn in enumerate(("v1", "v2", "v3")):
            for ci, c in enumerate("xyz"):
                sl[vn]["position"][c] = P[:, vi, ci]
                sl[vn]["normal"][c] = N[:, vi, ci]
            sl[vn]["texcoord"]["x"] = U[:, vi, 0]
            sl[vn]["texcoord"]["y"] = U[:, vi, 1]
        sl["mtl_index"] = mtl
        o += k
    return tris


def make_lights(directional=(), point=()):
    """directional: [(dir_towards_light, radiance)], point: [(position, radiance)]
    (Scene::AddDirectionalLight normalises, scene.cpp:347-351)."""
    lights = np.zeros(len(directional) + len(point), dtype=T.light)
    f32 = np.float32
    i = 0
    for d, rad in directional:
        d = np.asarray(d, f32)
        ln = f32(np.sqrt(f32(f32(f32(d[0] * d[0]) + f32(d[1] * d[1])) + f32(d[2] * d[2]))))
        for k, c in enumerate("xyz"):
            lights[i]["origin"][c] = f32(d[k] / ln)
            lights[i]["radiance"][c] = rad[k]
        lights[i]["type"] = 1
        i += 1
    for p, rad in point:
        for k, c in enumerate("xyz"):
            lights[i]["origin"][c] = p[k]
            lights[i]["radiance"][c] = rad[k]
        lights[i]["type"] = 0
        i += 1
    return lights


def checker_texture(size=64, cells=8, c0=(230, 230, 230), c1=(40, 60, 160)):
    """RGBA8 texels packed r | g<<8 | b<<16 | a<<24 (LoadSTB, image_loader.cpp:47-59)."""
    y, x = np.mgrid[0:size, 0:size]
    on = ((x * cells // size) + (y * cells // size)) % 2 == 0
    c0 = np.array(c0, np.uint32)
    c1 = np.array(c1, np.uint32)
    rgb = np.where(on[..., None], c0, c1)
    return (rgb[..., 0] | (rgb[..., 1] << 8) | (rgb[..., 2] << 16) | (np.uint32(255) << 24)).astype(np.uint32).ravel()


def gradient_env(width=64, height=32):
    """Small synthetic lat-long environment (float RGBA), used when the HDR asset
    is not wanted."""
    v = np.linspace(0, 1, height, dtype=np.float32)[:, None]
    u = np.linspace(0, 1, width, dtype=np.float32)[None, :]
    env = np.zeros((height, width, 4), np.float32)
    env[..., 0] = 0.3 + 0.7 * (1 - v) + 0.2 * np.sin(6.2831853 * u)
    env[..., 1] = 0.4 + 0.5 * (1 - v)
    env[..., 2] = 0.6 + 0.9 * (1 - v) * (0.5 + 0.5 * np.cos(6.2831853 * u))
    return np.maximum(env, 0).astype(np.float32)


# --------------------------------------------------------------------------
# Cornell box (assets/CornellBox.obj geometry: x,y in [-1,1], z in [0,2], Z-up,
# open towards -y where the default camera sits)
# --------------------------------------------------------------------------
CORNELL_MATERIALS = [
    dict(kd=(0.725, 0.71, 0.68)),                      # 0 white (floor, ceiling, back wall, boxes)
    dict(kd=(0.14, 0.45, 0.091)),                      # 1 rightWall (green)
    dict(kd=(0.63, 0.065, 0.05)),                      # 2 leftWall (red)
    dict(kd=(0.78, 0.78, 0.78), ke=(10, 10, 10)),      # 3 light
    # CornellBox_Dragon.mtl: dragon (rough metal) and teapot (mirror dielectric)
    dict(kd=(0.0, 1.0, 1.0), ks=(1.0, 0.75, 0.25), roughness=0.1, metalness=1.0),   # 4 dragon
    dict(kd=(1.0, 0.0, 0.0), ks=(1.0, 1.0, 1.0), roughness=0.0, metalness=0.0),     # 5 teapot
]


def cornell_shell():
    m = []
    m.append(quad((-1, -1, 0), (1, -1, 0), (1, 1, 0), (-1, 1, 0)) + (0,))          # floor, normal +z
    m.append(quad((-1, 1, 2), (1, 1, 2), (1, -1, 2), (-1, -1, 2)) + (0,))          # ceiling, normal -z
    m.append(quad((-1, 1, 0), (1, 1, 0), (1, 1, 2), (-1, 1, 2)) + (0,))            # back wall, normal -y
    m.append(quad((1, -1, 0), (1, -1, 2), (1, 1, 2), (1, 1, 0)) + (1,))            # right wall, normal -x
    m.append(quad((-1, -1, 0), (-1, 1, 0), (-1, 1, 2), (-1, -1, 2)) + (2,))        # left wall, normal +x
    m.append(quad((-0.25, 0.25, 1.99), (0.25, 0.25, 1.99), (0.25, -0.25, 1.99), (-0.25, -0.25, 1.99)) + (3,))
    return m


def cornell_blob(n_blob_tris=871_200, n_ball_tris=20_000, seed=1234):
    """Stand-in for BASELINE config 2 (CornellBox_Dragon.obj): the Cornell shell
    + a displaced high-poly blob with the `dragon` material + a tessellated
    sphere with the `teapot` material."""
    meshes = cornell_shell()
    n_lat = max(8, int(round(math.sqrt(n_blob_tris / 4.0))))
    n_lon = 2 * n_lat
    meshes.append(uv_sphere((0.25, 0.2, 0.62), 0.55, n_lat, n_lon, bump=0.35, seed=seed) + (4,))
    b_lat = max(6, int(round(math.sqrt(n_ball_tris / 4.0))))
    meshes.append(uv_sphere((-0.52, -0.3, 0.3), 0.3, b_lat, 2 * b_lat) + (5,))
    tris = to_triangles(meshes)
    mats = np.array([make_material(**m) for m in CORNELL_MATERIALS], dtype=T.packed_material)
    return tris, mats


def coverage_scene():
    """Small scene that touches every branch of the shading code: Lambert, mirror,
    rough GGX metal, textured albedo/roughness/emission, pass-through
    transparency, emissive quad, point + directional lights."""
    tex0 = checker_texture(64, 8)
    tex1 = checker_texture(32, 4, c0=(250, 250, 250), c1=(30, 30, 30))
    textures = np.zeros(2, dtype=T.texture)
    textures[0] = (0, 64, 64, 0)
    textures[1] = (64 * 64, 32, 32, 0)
    texture_data = np.concatenate([tex0, tex1]).astype(np.uint32)
    mats = [
        dict(kd=(0.725, 0.71, 0.68)),
        dict(kd=(0.14, 0.45, 0.091)),
        dict(kd=(0.63, 0.065, 0.05)),
        dict(kd=(0.78, 0.78, 0.78), ke=(10, 10, 10)),
        dict(kd=(0.0, 1.0, 1.0), ks=(1.0, 0.75, 0.25), roughness=0.1, metalness=1.0),
        dict(kd=(1.0, 0.0, 0.0), ks=(1.0, 1.0, 1.0), roughness=0.0, metalness=0.0),
        dict(kd=(0.6, 0.6, 0.6), kd_tex=0, roughness=0.4, r_tex=1),                     # 6 textured floor
        dict(kd=(0.2, 0.3, 0.9), ks=(0.5, 0.5, 0.5), roughness=0.3, metalness=0.5, ior=1.8),  # 7 mixed
        dict(kd=(0.9, 0.9, 0.9), transparency=0.0),                                     # 8 pass-through
        dict(kd=(0.5, 0.5, 0.5), ke=(2, 3, 4), e_tex=0, ks_tex=1, ks=(0.3, 0.3, 0.3), roughness=0.2),  # 9
        dict(kd=(0, 0, 0), ks=(0, 0, 0), ior=1.0),                                      # 10 zero-weight (NaN pdf)
        dict(kd=(0.8, 0.8, 0.2), transparency=0.9, t_tex=1, m_tex=1, metalness=0.7, roughness=0.05),   # 11
    ]
    meshes = cornell_shell()
    meshes[0] = quad((-1, -1, 0), (1, -1, 0), (1, 1, 0), (-1, 1, 0), uv_scale=3.0) + (6,)
    meshes.append(uv_sphere((0.35, 0.25, 0.45), 0.42, 24, 48, bump=0.3) + (4,))
    meshes.append(uv_sphere((-0.5, -0.25, 0.3), 0.3, 16, 32) + (5,))
    meshes.append(uv_sphere((-0.35, 0.45, 1.2), 0.28, 16, 32) + (7,))
    meshes.append(quad((-0.9, -0.6, 0.1), (-0.1, -0.6, 0.1), (-0.1, -0.6, 1.0), (-0.9, -0.6, 1.0)) + (8,))
    meshes.append(uv_sphere((0.6, -0.45, 1.3), 0.22, 12, 24) + (9,))
    meshes.append(box((0.55, -0.75, 0.0), (0.85, -0.45, 0.35)) + (10,))
    meshes.append(uv_sphere((0.0, -0.2, 1.55), 0.2, 12, 24) + (11,))
    tris = to_triangles(meshes)
    materials = np.array([make_material(**m) for m in mats], dtype=T.packed_material)
    lights = make_lights(directional=[((-0.6, -1.5, 3.5), (15.0, 10.0, 5.0))],
                         point=[((0.3, -0.7, 1.7), (1.5, 1.2, 0.9)), ((-0.7, 0.2, 0.4), (0.2, 0.5, 0.8))])
    return dict(triangles=tris, materials=materials, textures=textures, texture_data=texture_data, lights=lights)


def write_obj(path_obj, meshes, material_names, mtl_text):
    """Emit OBJ + MTL (used to exercise the scene loader on generated geometry)."""
    import os
    mtl_name = os.path.splitext(os.path.basename(path_obj))[0] + ".mtl"
    with open(os.path.join(os.path.dirname(path_obj), mtl_name), "w") as f:
        f.write(mtl_text)
    with open(path_obj, "w") as f:
        f.write("mtllib %s\n" % mtl_name)
        base = 1
        for P, N, U, mtl in meshes:
            f.write("o mesh%d\nusemtl %s\n" % (base, material_names[mtl]))
            n = len(P)
            for arr, tag in ((P.reshape(-1, 3), "v"), (U.reshape(-1, 2), "vt"), (N.reshape(-1, 3), "vn")):
                for row in arr:
                    f.write(tag + " " + " ".join(repr(float(x)) for x in row) + "\n")
            for t in range(n):
                a = base + 3 * t
                f.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (a, a, a, a + 1, a + 1, a + 1, a + 2, a + 2, a + 2))
            base += 3 * n


# --------------------------------------------------------------------------
# stand-ins for BASELINE configs 3-5
# --------------------------------------------------------------------------
def _instances(unit, centers, scales, rng, mtl_ids):
    """Vectorised instancing of a unit mesh: scale, rotate about z, translate."""
    P0, N0, U0 = unit
    k, m = len(centers), len(P0)
    ang = rng.uniform(0, 2 * np.pi, k)
    c, s = np.cos(ang), np.sin(ang)
    R = np.zeros((k, 3, 3))
    R[:, 0, 0], R[:, 0, 1], R[:, 1, 0], R[:, 1, 1], R[:, 2, 2] = c, -s, s, c, 1.0
    P = np.einsum("kij,mvj->kmvi", R, P0.astype(np.float64)) * scales[:, None, None, None] + centers[:, None, None, :]
    N = np.einsum("kij,mvj->kmvi", R, N0.astype(np.float64))
    U = np.broadcast_to(U0, (k,) + U0.shape)
    mt = np.repeat(mtl_ids, m)
    return (P.reshape(k * m, 3, 3).astype(np.float32), N.reshape(k * m, 3, 3).astype(np.float32),
            U.reshape(k * m, 3, 2).astype(np.float32), mt.astype(np.uint32))


def _triangles_with_ids(P, N, U, mt):
    """(n, 3, 3) positions and normals, (n, 3, 2) texture coordinates and n material ids as rt_types' 160-byte triangle records: per vertex 12 floats
    (position xyz_, texcoord xy__, normal xyz_), written as three block copies through a float view of the record array."""
    tris = np.zeros(len(P), dtype=T.triangle)
    v = tris.view(np.float32).reshape(len(P), 40)[:, :36].reshape(len(P), 3, 12)
    v[:, :, 0:3] = P
    v[:, :, 4:6] = U
    v[:, :, 8:11] = N
    tris["mtl_index"] = mt
    return tris


def random_materials(n, rng, n_textures=0):
    mats = []
    for i in range(n):
        kind = i % 4
        kd = tuple(rng.uniform(0.05, 0.9, 3))
        if kind == 0:
            m = dict(kd=kd)
        elif kind == 1:
            m = dict(kd=kd, ks=tuple(rng.uniform(0.5, 1.0, 3)), roughness=float(rng.uniform(0.05, 0.6)), metalness=1.0)
        elif kind == 2:
            m = dict(kd=kd, ks=(1, 1, 1), roughness=float(rng.uniform(0.0, 0.4)), metalness=0.0, ior=1.5)
        else:
            m = dict(kd=kd, ks=tuple(rng.uniform(0.2, 0.8, 3)), roughness=float(rng.uniform(0.2, 0.8)),
                     metalness=float(rng.uniform(0, 1)))
        if n_textures and i % 5 == 0:
            m["kd_tex"] = int(rng.randint(0, n_textures))
        mats.append(make_material(**m))
    return np.array(mats, dtype=T.packed_material)


def city_block(n_tris=2_800_000, seed=7, n_materials=120):
    """Stand-in for BASELINE config 4 (Amazon Lumberyard Bistro exterior, ~2.8 M triangles,
    100+ materials, some textured): a street grid of box 'buildings', tessellated 'props' and
    displaced 'foliage' blobs in front of the default camera."""
    rng = np.random.RandomState(seed)
    tex = [checker_texture(64, 8), checker_texture(64, 16, c0=(200, 180, 150), c1=(90, 60, 40)),
           checker_texture(32, 4, c0=(240, 240, 240), c1=(120, 160, 120))]
    textures = np.zeros(3, dtype=T.texture)
    off = 0
    for i, t in enumerate(tex):
        side = int(np.sqrt(len(t)))
        textures[i] = (off, side, side, 0)
        off += len(t)
    texture_data = np.concatenate(tex).astype(np.uint32)
    mats = random_materials(n_materials, rng, n_textures=3)
    parts = []
    # ground (two triangles, textured)
    g = quad((-30, -2, 0), (30, -2, 0), (30, 60, 0), (-30, 60, 0), uv_scale=40.0)
    parts.append((g[0], g[1], g[2], np.zeros(2, np.uint32)))
    # buildings: boxes on a jittered grid, 12 triangles each
    n_boxes = 3000
    bx = rng.uniform(-28, 28, n_boxes)
    by = rng.uniform(3, 58, n_boxes)
    keep = np.abs(bx) > 1.5                       # leave a street in front of the camera
    bx, by = bx[keep], by[keep]
    unit_box = box((-0.5, -0.5, 0.0), (0.5, 0.5, 1.0))
    for i in range(len(bx)):
        w, d, hgt = rng.uniform(0.6, 2.5), rng.uniform(0.6, 2.5), rng.uniform(1.0, 9.0)
        P = unit_box[0] * np.array([w, d, hgt], np.float32) + np.array([bx[i], by[i], 0], np.float32)
        parts.append((P, unit_box[1], unit_box[2] * 4.0, np.full(12, 1 + rng.randint(n_materials - 1), np.uint32)))
    used = sum(len(p[0]) for p in parts)
    # props: tessellated spheres (smooth) -- 40 % of the budget; foliage: displaced blobs -- the rest
    per_prop = 5000
    n_props = max(1, int(0.4 * (n_tris - used) / per_prop))
    lat = max(6, int(round(math.sqrt(per_prop / 4.0))))
    unit_s = uv_sphere((0, 0, 0), 1.0, lat, 2 * lat)
    centers = np.stack([rng.uniform(-25, 25, n_props), rng.uniform(1.5, 55, n_props), rng.uniform(0.2, 3.0, n_props)], 1)
    parts.append(_instances(unit_s, centers, rng.uniform(0.15, 0.6, n_props), rng, 1 + rng.randint(0, n_materials - 1, n_props)))
    used = sum(len(p[0]) for p in parts)
    per_blob = len(unit_s[0])
    unit_b = uv_sphere((0, 0, 0), 1.0, lat, 2 * lat, bump=0.5, seed=seed)
    n_blobs = max(1, (n_tris - used) // len(unit_b[0]))
    centers = np.stack([rng.uniform(-25, 25, n_blobs), rng.uniform(1.5, 55, n_blobs), rng.uniform(0.5, 6.0, n_blobs)], 1)
    parts.append(_instances(unit_b, centers, rng.uniform(0.3, 1.2, n_blobs), rng, 1 + rng.randint(0, n_materials - 1, n_blobs)))
    P = np.concatenate([p[0] for p in parts]); N = np.concatenate([p[1] for p in parts])
    U = np.concatenate([p[2] for p in parts]); mt = np.concatenate([p[3] for p in parts])
    tris = _triangles_with_ids(P, N, U, mt)
    return dict(triangles=tris, materials=mats, textures=textures, texture_data=texture_data)


def dense_foliage(n_tris=10_000_000, seed=11, n_materials=64):
    """Stand-in for BASELINE config 5 (San Miguel, ~10 M triangles, deep BVH, high depth
    complexity): thousands of overlapping displaced blobs inside a courtyard of walls."""
    rng = np.random.RandomState(seed)
    mats = random_materials(n_materials, rng)
    parts = []
    g = quad((-12, -2, 0), (12, -2, 0), (12, 30, 0), (-12, 30, 0))
    parts.append((g[0], g[1], g[2], np.zeros(2, np.uint32)))
    for wall in (quad((-12, 30, 0), (12, 30, 0), (12, 30, 8), (-12, 30, 8)),
                 quad((12, -2, 0), (12, -2, 8), (12, 30, 8), (12, 30, 0)),
                 quad((-12, -2, 0), (-12, 30, 0), (-12, 30, 8), (-12, -2, 8))):
        parts.append((wall[0], wall[1], wall[2], np.full(2, 1, np.uint32)))
    per = 4000
    lat = max(6, int(round(math.sqrt(per / 4.0))))
    unit_b = uv_sphere((0, 0, 0), 1.0, lat, 2 * lat, bump=0.6, seed=seed)
    n_blobs = max(1, n_tris // len(unit_b[0]))
    # nested clusters: blob centres drawn around ~200 cluster centres -> heavy overlap
    cl = np.stack([rng.uniform(-10, 10, 200), rng.uniform(1.0, 28, 200), rng.uniform(0.3, 5.0, 200)], 1)
    centers = cl[rng.randint(0, 200, n_blobs)] + rng.normal(0, 0.6, (n_blobs, 3))
    centers[:, 2] = np.abs(centers[:, 2]) + 0.1
    chunk = 250
    for a in range(0, n_blobs, chunk):
        b = min(n_blobs, a + chunk)
        parts.append(_instances(unit_b, centers[a:b], rng.uniform(0.08, 0.45, b - a), rng,
                                2 + rng.randint(0, n_materials - 2, b - a)))
    P = np.concatenate([p[0] for p in parts]); N = np.concatenate([p[1] for p in parts])
    U = np.concatenate([p[2] for p in parts]); mt = np.concatenate([p[3] for p in parts])
    return dict(triangles=_triangles_with_ids(P, N, U, mt), materials=mats,
                textures=np.zeros(0, T.texture), texture_data=np.zeros(0, np.uint32))


def shader_balls_obj(directory, tris_per_ball=20_000):
    """Stand-in for BASELINE config 3 (ShaderBalls.obj): writes ShaderBallsStandIn.obj next to a
    copy of the reference's ShaderBalls.mtl -- a 3x3 grid of tessellated spheres using mat00..mat22,
    the floor and the three emissive quads -- and returns the OBJ path (loaded through Scene)."""
    import os
    import shutil
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shutil.copy(os.path.join(here, "assets", "ShaderBalls.mtl"), os.path.join(directory, "ShaderBalls.mtl"))
    lat = max(6, int(round(math.sqrt(tris_per_ball / 4.0))))
    names = ["mat%d%d" % (r, c) for r in range(3) for c in range(3)] + ["floor", "light", "light_red", "light_blue"]
    meshes = []
    for r in range(3):
        for c in range(3):
            meshes.append(uv_sphere((-1.3 + 1.3 * c, 1.2 + 1.3 * r, 0.5), 0.5, lat, 2 * lat) + (3 * r + c,))
    meshes.append(quad((-6, -2, 0), (6, -2, 0), (6, 8, 0), (-6, 8, 0)) + (9,))
    meshes.append(quad((-1, 1.5, 3.0), (1, 1.5, 3.0), (1, 3.0, 3.0), (-1, 3.0, 3.0), normal=(0, 0, -1)) + (10,))
    meshes.append(quad((-3.5, 1.0, 0.2), (-3.5, 2.0, 0.2), (-3.5, 2.0, 1.5), (-3.5, 1.0, 1.5), normal=(1, 0, 0)) + (11,))
    meshes.append(quad((3.5, 2.0, 0.2), (3.5, 1.0, 0.2), (3.5, 1.0, 1.5), (3.5, 2.0, 1.5), normal=(-1, 0, 0)) + (12,))
    path = os.path.join(directory, "ShaderBallsStandIn.obj")
    mtl_name = "ShaderBalls.mtl"
    with open(path, "w") as f:
        f.write("mtllib %s\n" % mtl_name)
        base = 1
        for P, N, U, mtl in meshes:
            f.write("o mesh%d\nusemtl %s\n" % (base, names[mtl]))
            n = len(P)
            np.savetxt(f, P.reshape(-1, 3), fmt="v %.7g %.7g %.7g")
            np.savetxt(f, U.reshape(-1, 2), fmt="vt %.7g %.7g")
            np.savetxt(f, N.reshape(-1, 3), fmt="vn %.7g %.7g %.7g")
            idx = base + 3 * np.arange(n)[:, None] + np.arange(3)[None, :]
            np.savetxt(f, np.repeat(idx, 3, axis=1).reshape(n, 9)[:, [0, 1, 2, 3, 4, 5, 6, 7, 8]],
                       fmt="f %d/%d/%d %d/%d/%d %d/%d/%d")
            base += 3 * n
    return path


def blue_noise_tables(path=None):
    """(sobol_256spp_256d, scramblingTile, rankingTile) as int32 arrays from the packed asset
    (tools/make_blue_noise_asset.py)."""
    import os
    path = path or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets", "blue_noise",
                                "heitz2019_256spp_256d.bin")
    raw = np.fromfile(path, dtype=np.uint8)
    assert len(raw) == 65536 + 131072 + 131072
    return raw[:65536].astype(np.int32), raw[65536:65536 + 131072].astype(np.int32), raw[65536 + 131072:].astype(np.int32)
