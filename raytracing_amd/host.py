"""ctypes binding of raytracing_amd/librt_host.so -- the C++ host layer (Scene,
Bvh, HDR/TGA loaders, Render + HIPPathTraceIntegrator).  Plumbing only."""
import ctypes as C
import os
import numpy as np
from . import types as T
from .capi import rt_stats, RtError

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librt_host.so")
_lib = None

EXPORTS = [
    "rth_last_error", "rth_scene_load", "rth_scene_load_ex", "rth_scene_set_material_texture_indices", "rth_scene_set_emissive_nee",
    "rth_scene_emissive_nee", "rth_scene_from_arrays", "rth_scene_destroy",
    "rth_scene_add_directional_light", "rth_scene_add_point_light", "rth_scene_set_env_path",
    "rth_scene_set_env_image", "rth_scene_finalize", "rth_bvh_build", "rth_bvh_destroy", "rth_bvh_num_nodes",
    "rth_bvh_nodes", "rth_load_hdr", "rth_load_tga", "rth_load_png", "rth_loaded_image_data", "rth_default_camera",
    "rth_make_camera", "rth_render_create", "rth_render_destroy", "rth_render_set_camera",
    "rth_render_set_max_bounces", "rth_render_enable_white_furnace", "rth_render_set_sampler",
    "rth_render_enable_denoiser", "rth_render_set_resolve_every_frame", "rth_render_frame", "rth_render_samples",
    "rth_render_finish", "rth_render_local_rows", "rth_render_global_row", "rth_render_sample_count",
    "rth_render_read_radiance", "rth_render_read_resolved", "rth_render_stats", "rth_render_frame_handle",
    "rth_render_ctx_handle", "rth_render_num_nodes", "rth_render_nodes", "rth_render_set_aov", "rth_render_resolve",
    "rth_render_set_blue_noise_path", "rth_render_reserve_samples", "rth_scene_save_cache", "rth_load_jpeg",
    "rth_render_upload_gpu_data", "rth_render_setup_seconds", "rth_render_create_with_options",
]


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtError("librt_host.so is not built (run __graft_entry__.build())")
    lib = C.CDLL(LIB_PATH)
    vp, u32, f32, i32, cp = C.c_void_p, C.c_uint32, C.c_float, C.c_int, C.c_char_p
    sig = {
        "rth_last_error": (cp, []),
        "rth_scene_load": (vp, [cp, f32, i32]), "rth_scene_load_ex": (vp, [cp, f32, i32, u32]),
        "rth_scene_set_material_texture_indices": (i32, [vp, vp, u32]), "rth_scene_set_emissive_nee": (None, [vp, i32]),
        "rth_scene_emissive_nee": (i32, [vp]),
        "rth_scene_from_arrays": (vp, [vp, u32, vp, u32, vp, u32, vp, u32]),
        "rth_scene_destroy": (None, [vp]),
        "rth_scene_add_directional_light": (None, [vp] + [f32] * 6),
        "rth_scene_add_point_light": (None, [vp] + [f32] * 6),
        "rth_scene_set_env_path": (None, [vp, cp]), "rth_scene_set_env_image": (i32, [vp, vp, u32, u32]),
        "rth_scene_finalize": (i32, [vp]),
        "rth_bvh_build": (vp, [vp]), "rth_scene_save_cache": (i32, [vp, vp, C.c_char_p]), "rth_bvh_destroy": (None, [vp]), "rth_bvh_num_nodes": (u32, [vp]),
        "rth_bvh_nodes": (vp, [vp]),
        "rth_load_hdr": (i32, [cp, C.POINTER(u32), C.POINTER(u32)]),
        "rth_load_tga": (i32, [cp, C.POINTER(u32), C.POINTER(u32)]), "rth_loaded_image_data": (vp, []),
        "rth_load_png": (i32, [cp, C.POINTER(u32), C.POINTER(u32)]),
        "rth_load_jpeg": (i32, [cp, C.POINTER(u32), C.POINTER(u32)]),
        "rth_default_camera": (None, [u32, u32, vp]), "rth_make_camera": (None, [f32] * 9 + [vp]),
        "rth_render_create": (vp, [u32, u32, vp, i32, u32, u32, u32]), "rth_render_destroy": (None, [vp]),
        "rth_render_create_with_options": (vp, [u32, u32, vp, i32, u32, u32, u32, vp, u32]),
        "rth_render_set_camera": (i32, [vp, vp]), "rth_render_set_max_bounces": (i32, [vp, u32]),
        "rth_render_enable_white_furnace": (i32, [vp, i32]), "rth_render_set_sampler": (i32, [vp, i32]),
        "rth_render_enable_denoiser": (i32, [vp, i32]), "rth_render_set_resolve_every_frame": (i32, [vp, i32]),
        "rth_render_frame": (i32, [vp]), "rth_render_samples": (i32, [vp, u32]), "rth_render_reserve_samples": (i32, [vp, u32]), "rth_render_finish": (i32, [vp]),
        "rth_render_setup_seconds": (None, [vp, C.POINTER(C.c_double)]),
        "rth_render_local_rows": (u32, [vp]), "rth_render_global_row": (u32, [vp, u32]),
        "rth_render_sample_count": (u32, [vp]), "rth_render_read_radiance": (i32, [vp, vp]),
        "rth_render_read_resolved": (i32, [vp, vp]), "rth_render_stats": (i32, [vp, C.POINTER(rt_stats)]),
        "rth_render_frame_handle": (vp, [vp]), "rth_render_ctx_handle": (vp, [vp]), "rth_render_upload_gpu_data": (i32, [vp]),
        "rth_render_num_nodes": (u32, [vp]), "rth_render_nodes": (vp, [vp]),
        "rth_render_set_aov": (i32, [vp, i32]), "rth_render_resolve": (i32, [vp, vp]),
        "rth_render_set_blue_noise_path": (i32, [vp, cp]),
    }
    for name in ("triangles", "materials", "textures", "texture_data", "lights", "emissive", "material_texture_indices"):
        sig["rth_scene_num_" + name] = (u32, [vp])
        sig["rth_scene_" + name] = (vp, [vp])
    sig["rth_scene_env_width"] = (u32, [vp])
    sig["rth_scene_env_height"] = (u32, [vp])
    sig["rth_scene_env_data"] = (vp, [vp])
    for k, (res, args) in sig.items():
        f = getattr(lib, k)
        f.restype, f.argtypes = res, args
    _lib = lib
    return lib


def _err(lib):
    return RtError(lib.rth_last_error().decode())


def _arr(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


def default_camera(width, height):
    cam = np.zeros((), dtype=T.camera)
    load().rth_default_camera(width, height, cam.ctypes.data)
    return cam


def load_hdr(path):
    lib = load()
    w, h = C.c_uint32(), C.c_uint32()
    if lib.rth_load_hdr(path.encode(), C.byref(w), C.byref(h)):
        raise RtError("LoadHDR failed: " + path)
    return _arr(lib.rth_loaded_image_data(), w.value * h.value * 4, np.float32).reshape(h.value, w.value, 4)


def load_tga(path):
    lib = load()
    w, h = C.c_uint32(), C.c_uint32()
    if lib.rth_load_tga(path.encode(), C.byref(w), C.byref(h)):
        raise RtError("LoadTGA failed: " + path)
    return _arr(lib.rth_loaded_image_data(), w.value * h.value, np.uint32).reshape(h.value, w.value)


def load_jpeg(path):
    lib = load()
    w, h = C.c_uint32(), C.c_uint32()
    if lib.rth_load_jpeg(path.encode(), C.byref(w), C.byref(h)):
        raise RtError("LoadJPEG failed: " + path)
    return _arr(lib.rth_loaded_image_data(), w.value * h.value, np.uint32).reshape(h.value, w.value)


def load_png(path):
    lib = load()
    w, h = C.c_uint32(), C.c_uint32()
    if lib.rth_load_png(path.encode(), C.byref(w), C.byref(h)):
        raise RtError("LoadPNG failed: " + path)
    return _arr(lib.rth_loaded_image_data(), w.value * h.value, np.uint32).reshape(h.value, w.value)


class Scene:
    """rt::Scene (reference surface: src/scene/scene.hpp:34-67)."""

    _GETTERS = (("triangles", T.triangle), ("materials", T.packed_material), ("textures", T.texture),
                ("texture_data", np.uint32), ("lights", T.light), ("emissive", np.uint32))

    def __init__(self, path=None, scale=1.0, flip_yz=False, arrays=None, wide_texture_indices=False, emissive_nee=False):
        """wide_texture_indices / emissive_nee: this repository's opt-in extensions (rt::Scene::Options)"""
        self.lib = load()
        self.bvh = None
        if arrays is not None:
            a = {k: np.ascontiguousarray(v) for k, v in arrays.items()}
            p = lambda x: x.ctypes.data if x.size else None
            tex = a.get("textures", np.zeros(0, T.texture))
            td = a.get("texture_data", np.zeros(0, np.uint32))
            self.handle = self.lib.rth_scene_from_arrays(p(a["triangles"]), len(a["triangles"]), p(a["materials"]),
                                                         len(a["materials"]), p(tex), len(tex), p(td), len(td))
            if self.handle and a.get("material_texture_indices") is not None:
                t16 = np.ascontiguousarray(a["material_texture_indices"], np.uint16)
                if self.lib.rth_scene_set_material_texture_indices(self.handle, t16.ctypes.data, t16.size):
                    raise _err(self.lib)
            if self.handle and emissive_nee:
                self.lib.rth_scene_set_emissive_nee(self.handle, 1)
        else:
            self.handle = self.lib.rth_scene_load_ex(path.encode(), scale, int(flip_yz),
                                                     (1 if wide_texture_indices else 0) | (2 if emissive_nee else 0))
        if not self.handle:
            raise _err(self.lib)

    def add_directional_light(self, direction, radiance):
        self.lib.rth_scene_add_directional_light(self.handle, *direction, *radiance)

    def add_point_light(self, origin, radiance):
        self.lib.rth_scene_add_point_light(self.handle, *origin, *radiance)

    def set_env_path(self, path):
        self.lib.rth_scene_set_env_path(self.handle, path.encode())

    def set_env_image(self, rgba):
        rgba = np.ascontiguousarray(rgba, np.float32)
        self.lib.rth_scene_set_env_image(self.handle, rgba.ctypes.data, rgba.shape[1], rgba.shape[0])

    def build_bvh(self):
        """Bvh::BuildCPU on this scene's triangles (reorders them)."""
        h = self.lib.rth_bvh_build(self.handle)
        if not h:
            raise _err(self.lib)
        self.bvh = h
        return _arr(self.lib.rth_bvh_nodes(h), self.lib.rth_bvh_num_nodes(h), T.bvh_node)

    def finalize(self):
        if self.lib.rth_scene_finalize(self.handle):
            raise _err(self.lib)

    def save_cache(self, path):
        """Binary scene cache: reordered triangles + BVH nodes + materials + textures; Scene(path) loads it."""
        if not self.bvh:
            self.build_bvh()
        if self.lib.rth_scene_save_cache(self.handle, self.bvh, path.encode()):
            raise _err(self.lib)

    def arrays(self):
        out = {}
        for name, dt in self._GETTERS:
            n = getattr(self.lib, "rth_scene_num_" + name)(self.handle)
            out[name] = _arr(getattr(self.lib, "rth_scene_" + name)(self.handle), n, dt)
        w, h = self.lib.rth_scene_env_width(self.handle), self.lib.rth_scene_env_height(self.handle)
        out["env"] = _arr(self.lib.rth_scene_env_data(self.handle), w * h * 4, np.float32).reshape(h, w, 4)
        if self.bvh:
            out["nodes"] = _arr(self.lib.rth_bvh_nodes(self.bvh), self.lib.rth_bvh_num_nodes(self.bvh), T.bvh_node)
        # opt-in extensions, present only when used (capi.Context.upload_scene / tests/_oracle.py read the same keys)
        n16 = self.lib.rth_scene_num_material_texture_indices(self.handle)
        if n16:
            out["material_texture_indices"] = _arr(self.lib.rth_scene_material_texture_indices(self.handle), n16, np.uint16).reshape(-1, 6)
        if self.lib.rth_scene_emissive_nee(self.handle):
            out["flags"] = 1
        return out

    def close(self):
        if self.bvh:
            self.lib.rth_bvh_destroy(self.bvh)
            self.bvh = None
        if self.handle:
            self.lib.rth_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Render:
    """rt::Render: headless Render(width, height, scene) -> RenderFrame()."""

    def __init__(self, width, height, scene, device=0, tile_rank=0, tile_count=1, band_height=8, ctx_options=()):
        """ctx_options: (rt_ctx_option, value) pairs set on the context BEFORE the scene is uploaded (a rank that will take another rank's folds uploads
        without a shadow tree and an adaptation of its own: ((2, 0), (4, 0)))"""
        self.lib = load()
        self.scene = scene
        self.width, self.height = width, height
        if ctx_options:
            flat = np.asarray([v for pair in ctx_options for v in pair], np.uint32)
            self.handle = self.lib.rth_render_create_with_options(width, height, scene.handle, device, tile_rank, tile_count, band_height, flat.ctypes.data, len(ctx_options))
        else:
            self.handle = self.lib.rth_render_create(width, height, scene.handle, device, tile_rank, tile_count,
                                                     band_height)
        if not self.handle:
            raise _err(self.lib)
        self.local_rows = self.lib.rth_render_local_rows(self.handle)

    def _c(self, rc):
        if rc:
            raise _err(self.lib)

    def set_camera(self, cam):
        cam = np.ascontiguousarray(cam)
        self._c(self.lib.rth_render_set_camera(self.handle, cam.ctypes.data))

    def set_max_bounces(self, b): self._c(self.lib.rth_render_set_max_bounces(self.handle, b))
    def enable_white_furnace(self, e): self._c(self.lib.rth_render_enable_white_furnace(self.handle, int(e)))
    def set_blue_noise(self, e, table_path=None):
        path = table_path or os.path.join(os.path.dirname(_HERE), "assets", "blue_noise", "heitz2019_256spp_256d.bin")
        self.lib.rth_render_set_blue_noise_path(self.handle, path.encode())
        self._c(self.lib.rth_render_set_sampler(self.handle, int(e)))
    def enable_denoiser(self, e): self._c(self.lib.rth_render_enable_denoiser(self.handle, int(e)))
    def set_aov(self, aov): self._c(self.lib.rth_render_set_aov(self.handle, int(aov)))

    def set_wide_bvh(self, mode):
        """RT_CTX_OPT_WIDE_BVH (1 = SAH-optimal frontier per wide record, the default; 2 = two BVH2 levels per record; 0 = none),
        then the scene is uploaded again: A/B runs and tools."""
        from . import capi
        if capi.load().rt_ctx_set_option(self.lib.rth_render_ctx_handle(self.handle), 1, mode):
            raise _err(self.lib)
        self._c(self.lib.rth_render_upload_gpu_data(self.handle))

    def set_ctx_option(self, option, value, upload=True):
        """rt_ctx_set_option on this Render's context (then the scene is uploaded again, as the options take effect there)"""
        from . import capi
        if capi.load().rt_ctx_set_option(self.lib.rth_render_ctx_handle(self.handle), option, value):
            raise _err(self.lib)
        if upload:
            self._c(self.lib.rth_render_upload_gpu_data(self.handle))

    def set_shadow_tree(self, mode, upload=True):
        """RT_CTX_OPT_SHADOW_TREE: 1 (default) = the backend's own tree for shadow rays where it measures cheaper, 2 = always,
        3 = always, surface-area metric, 0 = shadow rays share the closest-hit tree.  Bit-identical results for every value."""
        self.set_ctx_option(2, mode, upload)

    def set_closest_tree(self, mode, upload=True):
        """RT_CTX_OPT_CLOSEST_TREE: 0 (default) = the reference's topology and order (bit-identical); 1 / 2 = TOLERANCE mode,
        an own tree for closest-hit rays where it measures cheaper / always."""
        self.set_ctx_option(3, mode, upload)

    def set_adaptive_fold(self, mode, upload=True):
        """RT_CTX_OPT_ADAPTIVE_FOLD: bit 0 = the first frame probes its own rays and both 4-wide trees are folded again for them (exact; the
        report line appears in tree_report() once the fold is adopted), bit 1 = the frame waits for it, bit 2 = small trees too."""
        self.set_ctx_option(4, mode, upload)

    def tree_report(self):
        from . import capi
        return capi.load().rt_scene_tree_report(self.lib.rth_render_ctx_handle(self.handle)).decode()

    def resolve_now(self):
        out = np.zeros((self.local_rows, self.width, 4), np.float32)
        self._c(self.lib.rth_render_resolve(self.handle, out.ctypes.data))
        return out
    def set_resolve_every_frame(self, e): self._c(self.lib.rth_render_set_resolve_every_frame(self.handle, int(e)))
    def setup_seconds(self):
        """what the constructor spent: BVH build (or adoption of a cached tree), Scene::Finalize, the integrator's frame, UploadGPUData"""
        out = (C.c_double * 4)()
        self.lib.rth_render_setup_seconds(self.handle, out)
        return dict(bvh_build=round(out[0], 3), finalize=round(out[1], 3), frame=round(out[2], 3), upload=round(out[3], 3))

    def render_frame(self): self._c(self.lib.rth_render_frame(self.handle))
    def render_samples(self, n): self._c(self.lib.rth_render_samples(self.handle, n))

    def reserve_samples(self, n):
        """Size the device's per-path buffers for render_samples(n); returns the samples traced together."""
        r = self.lib.rth_render_reserve_samples(self.handle, n)
        if r < 0:
            self._c(1)
        return r
    def finish(self): self._c(self.lib.rth_render_finish(self.handle))
    def sample_count(self): return self.lib.rth_render_sample_count(self.handle)

    def global_rows(self):
        return np.array([self.lib.rth_render_global_row(self.handle, r) for r in range(self.local_rows)], np.int64)

    def radiance(self):
        out = np.zeros((self.local_rows, self.width, 4), np.float32)
        self._c(self.lib.rth_render_read_radiance(self.handle, out.ctypes.data))
        return out

    def resolved(self):
        out = np.zeros((self.local_rows, self.width, 4), np.float32)
        self._c(self.lib.rth_render_read_resolved(self.handle, out.ctypes.data))
        return out

    def stats(self):
        st = rt_stats()
        self._c(self.lib.rth_render_stats(self.handle, C.byref(st)))
        return st

    def scene_arrays(self):
        """Scene + BVH arrays exactly as uploaded (triangles in BVH order)."""
        out = self.scene.arrays()
        out["nodes"] = _arr(self.lib.rth_render_nodes(self.handle), self.lib.rth_render_num_nodes(self.handle),
                            T.bvh_node)
        return out

    def radiance_device_ptr(self):
        from . import capi
        return capi.load().rt_frame_radiance_device_ptr(self.lib.rth_render_frame_handle(self.handle))

    def close(self):
        if self.handle:
            self.lib.rth_render_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
