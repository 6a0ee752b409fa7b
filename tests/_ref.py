"""ctypes binding of oracle/_ref/libref.so -- the reference's own kernels and
host producers compiled for x86-64 (oracle/Makefile).  TEST INFRASTRUCTURE."""
import ctypes as C
import os
import numpy as np
from raytracing_amd import types as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lib_path(libm=False):
    return os.path.join(ROOT, "oracle", "_ref", "libref_libm.so" if libm else "libref.so")


def available(libm=False):
    return os.path.exists(lib_path(libm))


_libs = {}


def load(libm=False):
    if libm in _libs:
        return _libs[libm]
    lib = C.CDLL(lib_path(libm))
    vp, u32, f32, cp = C.c_void_p, C.c_uint32, C.c_float, C.c_char_p
    sig = {
        "refh_scene_load": (vp, [cp, f32, C.c_int]), "refh_scene_destroy": (None, [vp]),
        "refh_add_directional_light": (None, [vp] + [f32] * 6), "refh_add_point_light": (None, [vp] + [f32] * 6),
        "refh_build_and_finalize": (None, [vp]),
        "refh_scene_info": (None, [vp, vp]),
        "refh_bvh_build": (u32, [vp, u32]), "refh_bvh_nodes": (None, [vp]),
        "refh_load_hdr": (C.c_int, [cp, C.POINTER(u32), C.POINTER(u32)]), "refh_loaded_image_data": (vp, []),
        "refh_load_stb": (C.c_int, [cp, C.POINTER(u32), C.POINTER(u32)]),
        "ref_create": (vp, [u32, u32, C.c_int, C.c_int]), "ref_destroy": (None, [vp]),
        "ref_upload": (None, [vp, vp, u32, vp, u32, vp, u32, vp, u32, vp, u32, vp, u32, vp, u32, vp, u32, u32]),
        "ref_set_camera": (None, [vp, vp]), "ref_set_max_bounces": (None, [vp, u32]),
        "ref_request_reset": (None, [vp]), "ref_integrate": (None, [vp]),
        "ref_resolve": (vp, [vp]), "ref_radiance": (vp, [vp]), "ref_sample_count": (u32, [vp]),
        "ref_ray_totals": (None, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "ref_last_counts": (None, [vp, vp, vp, u32]), "ref_buffer": (vp, [vp, cp]),
        "ref_stage_reset": (None, [vp]), "ref_stage_generate_rays": (None, [vp]),
        "ref_stage_intersect": (None, [vp, u32]), "ref_stage_shade_miss": (None, [vp, u32]),
        "ref_stage_clear_counters": (None, [vp, u32]), "ref_stage_shade_hits": (None, [vp, u32]),
        "ref_stage_intersect_shadow": (None, [vp]), "ref_stage_accumulate": (None, [vp]),
        "ref_stage_advance": (None, [vp]),
        "ref_set_blue_noise_tables": (None, [vp, vp, vp, vp]), "ref_set_sampler": (None, [vp, C.c_int]),
        "ref_enable_denoiser": (None, [vp, C.c_int]), "ref_set_aov": (None, [vp, u32]),
    }
    for name in ("triangles", "nodes", "materials", "textures", "texture_data", "lights", "emissive"):
        sig["refh_num_" + name] = (u32, [vp])
        sig["refh_" + name] = (vp, [vp])
    sig["refh_env_width"] = (u32, [vp]); sig["refh_env_height"] = (u32, [vp]); sig["refh_env_data"] = (vp, [vp])
    for k, (res, args) in sig.items():
        f = getattr(lib, k)
        f.restype, f.argtypes = res, args
    _libs[libm] = lib
    return lib


def _arr(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


def load_scene(path, scale=1.0, flip_yz=False, dir_lights=(), point_lights=(), libm=False):
    """Reference pipeline main.cpp:56-58 + render.cpp:61-67 -> dict of numpy arrays.
    Must run with CWD = a directory holding assets/ibl/CGSkies_0036_free.hdr."""
    lib = load(libm)
    h = lib.refh_scene_load(path.encode(), scale, int(flip_yz))
    if not h:
        raise RuntimeError("reference Scene failed to load " + path)
    for d, r in dir_lights:
        lib.refh_add_directional_light(h, *d, *r)
    for p, r in point_lights:
        lib.refh_add_point_light(h, *p, *r)
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        lib.refh_build_and_finalize(h)
    finally:
        os.chdir(cwd)
    out = {}
    for name, dt in (("triangles", T.triangle), ("nodes", T.bvh_node), ("materials", T.packed_material),
                     ("textures", T.texture), ("texture_data", np.uint32), ("lights", T.light),
                     ("emissive", np.uint32)):
        n = getattr(lib, "refh_num_" + name)(h)
        out[name] = _arr(getattr(lib, "refh_" + name)(h), n, dt)
    w, hh = lib.refh_env_width(h), lib.refh_env_height(h)
    out["env"] = _arr(lib.refh_env_data(h), w * hh * 4, np.float32).reshape(hh, w, 4)
    si = np.zeros((), dtype=T.scene_info)
    lib.refh_scene_info(h, si.ctypes.data)
    out["scene_info"] = si
    lib.refh_scene_destroy(h)
    return out


def bvh_build(tris, libm=False):
    lib = load(libm)
    tris = np.ascontiguousarray(tris.copy())
    n = lib.refh_bvh_build(tris.ctypes.data, len(tris))
    nodes = np.zeros(n, dtype=T.bvh_node)
    lib.refh_bvh_nodes(nodes.ctypes.data)
    return tris, nodes


def load_hdr(path, libm=False):
    lib = load(libm)
    w, h = C.c_uint32(), C.c_uint32()
    if not lib.refh_load_hdr(path.encode(), C.byref(w), C.byref(h)):
        raise RuntimeError("LoadHDR failed")
    return _arr(lib.refh_loaded_image_data(), w.value * h.value * 4, np.float32).reshape(h.value, w.value, 4)


def load_stb(path, libm=False):
    """The reference's LoadSTB (stb_image) -> packed RGBA8 texels."""
    lib = load(libm)
    w, h = C.c_uint32(), C.c_uint32()
    if not lib.refh_load_stb(path.encode(), C.byref(w), C.byref(h)):
        raise RuntimeError("LoadSTB failed")
    return _arr(lib.refh_loaded_image_data(), w.value * h.value, np.uint32).reshape(h.value, w.value)


class RefIntegrator:
    def __init__(self, width, height, scene, furnace=False, threads=0, libm=False):
        self.lib = load(libm)
        self.w, self.h = width, height
        self.handle = self.lib.ref_create(width, height, int(furnace), threads)
        s = {k: np.ascontiguousarray(v) for k, v in scene.items()}
        self._keep = s
        p = lambda a: a.ctypes.data if a.size else None
        env = s["env"]
        self.lib.ref_upload(self.handle, p(s["triangles"]), len(s["triangles"]), p(s["nodes"]), len(s["nodes"]),
                            p(s["materials"]), len(s["materials"]), p(s["textures"]), len(s["textures"]),
                            p(s["texture_data"]), len(s["texture_data"]), p(s["lights"]), len(s["lights"]),
                            p(s["emissive"]), len(s["emissive"]), p(env), env.shape[1], env.shape[0])

    def set_camera(self, cam):
        self._cam = np.ascontiguousarray(cam)
        self.lib.ref_set_camera(self.handle, self._cam.ctypes.data)

    def set_max_bounces(self, b):
        self.lib.ref_set_max_bounces(self.handle, b)

    def set_blue_noise(self, enable, tables=None):
        if tables is not None:
            self._bn = [np.ascontiguousarray(t, np.int32) for t in tables]
            self.lib.ref_set_blue_noise_tables(self.handle, *[t.ctypes.data for t in self._bn])
        self.lib.ref_set_sampler(self.handle, int(enable))

    def enable_denoiser(self, e):
        self.lib.ref_enable_denoiser(self.handle, int(e))

    def set_aov(self, aov):
        self.lib.ref_set_aov(self.handle, aov)

    def integrate(self, n=1):
        for _ in range(n):
            self.lib.ref_integrate(self.handle)

    def radiance(self):
        n = self.w * self.h
        return _arr(self.lib.ref_radiance(self.handle), n * 4, np.float32).reshape(self.h, self.w, 4)

    def resolve(self):
        n = self.w * self.h
        return _arr(self.lib.ref_resolve(self.handle), n * 4, np.float32).reshape(self.h, self.w, 4)

    def sample_count(self):
        return self.lib.ref_sample_count(self.handle)

    def ray_totals(self):
        a, b = C.c_uint64(), C.c_uint64()
        self.lib.ref_ray_totals(self.handle, C.byref(a), C.byref(b))
        return a.value, b.value

    def last_counts(self, n):
        a = np.zeros(n, np.uint32); b = np.zeros(n, np.uint32)
        self.lib.ref_last_counts(self.handle, a.ctypes.data, b.ctypes.data, n)
        return a, b

    def buffer(self, name, dtype, count):
        return _arr(self.lib.ref_buffer(self.handle, name.encode()), count, dtype)

    def __del__(self):
        try:
            self.lib.ref_destroy(self.handle)
        except Exception:
            pass


def _stage(self, name, *args):
    getattr(self.lib, "ref_stage_" + name)(self.handle, *args)


RefIntegrator.stage = _stage
