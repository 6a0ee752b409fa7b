"""ctypes binding of oracle/liboracle.so (oracle/oracle.c, the plain-C
restatement).  TEST INFRASTRUCTURE: imported only by tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke()."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "liboracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])


def load():
    global _lib
    if _lib is not None:
        return _lib
    src = os.path.join(ROOT, "oracle", "oracle.c")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        build()
    lib = C.CDLL(LIB)
    vp, u32, f32, cp = C.c_void_p, C.c_uint32, C.c_float, C.c_char_p
    sig = {
        "orc_create": (vp, [u32, u32, C.c_int]), "orc_destroy": (None, [vp]),
        "orc_upload": (None, [vp, vp, u32, vp, u32, vp, u32, vp, u32, vp, u32, vp, u32, vp, u32, vp, u32, u32]),
        "orc_set_extensions": (None, [vp, vp, u32]),
        "orc_set_camera": (None, [vp, vp]), "orc_set_max_bounces": (None, [vp, u32]),
        "orc_request_reset": (None, [vp]), "orc_integrate": (None, [vp]),
        "orc_resolve": (vp, [vp]), "orc_radiance": (vp, [vp]), "orc_sample_count": (u32, [vp]),
        "orc_ray_totals": (None, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "orc_last_counts": (None, [vp, vp, vp, u32]), "orc_buffer": (vp, [vp, cp]), "orc_stats": (None, [vp, vp]),
        "orc_stage_reset": (None, [vp]), "orc_stage_generate_rays": (None, [vp]),
        "orc_stage_intersect": (None, [vp, u32]), "orc_stage_shade_miss": (None, [vp, u32]),
        "orc_stage_clear_counters": (None, [vp, u32]), "orc_stage_shade_hits": (None, [vp, u32]),
        "orc_stage_intersect_shadow": (None, [vp]), "orc_stage_accumulate": (None, [vp]),
        "orc_stage_advance": (None, [vp]),
        "orc_set_blue_noise_tables": (None, [vp, vp, vp]), "orc_set_sampler": (None, [vp, C.c_int]),
        "orc_enable_denoiser": (None, [vp, C.c_int]), "orc_set_aov": (None, [vp, u32]),
        "orc_wang_hash": (u32, [u32]), "orc_sample_random": (f32, [u32] * 5),
        "orc_tanf": (f32, [f32]), "orc_sinf": (f32, [f32]), "orc_cosf": (f32, [f32]),
        "orc_powf": (f32, [f32, f32]), "orc_atan2f": (f32, [f32, f32]), "orc_acosf": (f32, [f32]),
        "orc_wide_trace": (C.c_int, [vp, vp, u32, u32, vp, u32, C.c_int, vp, vp, vp]),
        "orc_wide_trace_events": (C.c_int, [vp, vp, u32, u32, vp, u32, C.c_int, vp, vp, vp, vp, u32, vp]),
    }
    for k, (res, args) in sig.items():
        f = getattr(lib, k)
        f.restype, f.argtypes = res, args
    _lib = lib
    return lib


def _arr(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


STAT_NAMES = ("closest_nodes", "closest_tris", "shadow_nodes", "shadow_tris", "escaped", "hits", "emissive",
              "texels", "unoccluded", "outgoing")


class Oracle:
    """Same surface as tests/_ref.RefIntegrator."""

    def __init__(self, width, height, scene, furnace=False):
        self.lib = load()
        self.w, self.h = width, height
        self.handle = self.lib.orc_create(width, height, int(furnace))
        s = {k: np.ascontiguousarray(v) for k, v in scene.items() if k not in ("flags",)}
        p = lambda a: a.ctypes.data if a.size else None
        env = s["env"]
        self.lib.orc_upload(self.handle, p(s["triangles"]), len(s["triangles"]), p(s["nodes"]), len(s["nodes"]),
                            p(s["materials"]), len(s["materials"]), p(s["textures"]), len(s["textures"]),
                            p(s["texture_data"]), len(s["texture_data"]), p(s["lights"]), len(s["lights"]),
                            p(s["emissive"]), len(s["emissive"]), p(env), env.shape[1], env.shape[0])
        # this repository's opt-in extensions (rt_scene_desc::material_texture_indices / flags)
        tex16 = s.get("material_texture_indices")
        flags = int(scene.get("flags", 0))
        if tex16 is not None or flags:
            if tex16 is not None:
                tex16 = np.ascontiguousarray(tex16, np.uint16)
                assert tex16.size == 6 * len(s["materials"])
            self.lib.orc_set_extensions(self.handle, tex16.ctypes.data if tex16 is not None else None, flags)

    def set_camera(self, cam):
        self._cam = np.ascontiguousarray(cam)
        self.lib.orc_set_camera(self.handle, self._cam.ctypes.data)

    def set_max_bounces(self, b):
        self.lib.orc_set_max_bounces(self.handle, b)

    def set_blue_noise(self, enable, tables=None):
        if tables is not None:
            self._bn = [np.ascontiguousarray(t, np.int32) for t in tables]
            self.lib.orc_set_blue_noise_tables(*[t.ctypes.data for t in self._bn])
        self.lib.orc_set_sampler(self.handle, int(enable))

    def enable_denoiser(self, e):
        self.lib.orc_enable_denoiser(self.handle, int(e))

    def set_aov(self, aov):
        self.lib.orc_set_aov(self.handle, aov)

    def integrate(self, n=1):
        for _ in range(n):
            self.lib.orc_integrate(self.handle)

    def radiance(self):
        return _arr(self.lib.orc_radiance(self.handle), self.w * self.h * 4, np.float32).reshape(self.h, self.w, 4)

    def resolve(self):
        return _arr(self.lib.orc_resolve(self.handle), self.w * self.h * 4, np.float32).reshape(self.h, self.w, 4)

    def sample_count(self):
        return self.lib.orc_sample_count(self.handle)

    def ray_totals(self):
        a, b = C.c_uint64(), C.c_uint64()
        self.lib.orc_ray_totals(self.handle, C.byref(a), C.byref(b))
        return a.value, b.value

    def last_counts(self, n):
        a = np.zeros(n, np.uint32); b = np.zeros(n, np.uint32)
        self.lib.orc_last_counts(self.handle, a.ctypes.data, b.ctypes.data, n)
        return a, b

    def stats(self):
        st = np.zeros(10, np.uint64)
        self.lib.orc_stats(self.handle, st.ctypes.data)
        return dict(zip(STAT_NAMES, (int(x) for x in st)))

    def buffer(self, name, dtype, count):
        return _arr(self.lib.orc_buffer(self.handle, name.encode()), count, dtype)

    def stage(self, name, *args):
        getattr(self.lib, "orc_stage_" + name)(self.handle, *args)

    WIDE_COUNTERS = ("rays", "wide_visits", "leaf_arrivals", "leaf_box_fails", "triangle_tests", "pushes", "culled_pops",
                     "deepest_stack", "rays_left_to_bvh2", "slots_passed")

    def wide_trace(self, wide_records, entry_ref, rays, shadow, counters=None, direct=False, ordered_shadow=False, by_distance=False):
        """k_trace_w4's walk restated on the CPU (oracle.c: orc_wide_trace) over the records of rt_debug_wide_bvh, for the
        rays given (records of types.ray).  Returns hits (types.hit) or, shadow, the shadow-hit words; adds to `counters`
        (np.uint64[10], WIDE_COUNTERS) when given."""
        rays = np.ascontiguousarray(rays)
        wide = np.ascontiguousarray(wide_records)
        cnt = counters if counters is not None else np.zeros(10, np.uint64)
        hits = np.zeros(len(rays), np.dtype([("bc", "<f4", 2), ("primitive_id", "<u4"), ("t", "<f4")]))
        sh = np.zeros(len(rays), np.uint32)
        rc = self.lib.orc_wide_trace(self.handle, wide.ctypes.data if len(wide) else None, len(wide), entry_ref,
                                     rays.ctypes.data if len(rays) else None, len(rays), int(bool(shadow)) | (2 if direct else 0) | (4 if ordered_shadow else 0) | (8 if by_distance else 0),
                                     hits.ctypes.data, sh.ctypes.data, cnt.ctypes.data)
        if rc != 0:
            raise RuntimeError("orc_wide_trace failed: %d" % rc)
        return sh if shadow else hits

    def nwide_stats(self, width, rays, shadow, nodes=None):
        """orc_nwide_stats: the SAH-optimal `width`-wide fold of a BVH2 (default: the oracle's own = the reference's) walked with exact
        boxes.  Returns dict(rays, visits, leaf_arrivals, triangle_tests, slots_tested, records)."""
        rays = np.ascontiguousarray(rays)
        cnt = np.zeros(6, np.uint64)
        self.lib.orc_nwide_stats.restype = C.c_int
        self.lib.orc_nwide_stats.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
        nd = np.ascontiguousarray(nodes) if nodes is not None else None
        rc = self.lib.orc_nwide_stats(self.handle, width, nd.ctypes.data if nd is not None else None, len(nd) if nd is not None else 0,
                                      rays.ctypes.data if len(rays) else None, len(rays), int(bool(shadow)), cnt.ctypes.data)
        if rc != 0:
            raise RuntimeError("orc_nwide_stats failed: %d" % rc)
        return dict(zip(("rays", "visits", "leaf_arrivals", "triangle_tests", "slots_tested", "records"), (int(x) for x in cnt)))

    def wide_trace_events(self, wide_records, entry_ref, rays, shadow, stride=192, direct=False):
        """The step sequence of every ray's walk: (events uint8[n, stride] of b'N' / b'L' / b'T', lengths uint32[n])."""
        rays = np.ascontiguousarray(rays)
        wide = np.ascontiguousarray(wide_records)
        cnt = np.zeros(10, np.uint64)
        hits = np.zeros(len(rays), np.dtype([("bc", "<f4", 2), ("primitive_id", "<u4"), ("t", "<f4")]))
        sh = np.zeros(len(rays), np.uint32)
        ev = np.zeros((len(rays), stride), np.uint8)
        ln = np.zeros(len(rays), np.uint32)
        rc = self.lib.orc_wide_trace_events(self.handle, wide.ctypes.data if len(wide) else None, len(wide), entry_ref,
                                            rays.ctypes.data if len(rays) else None, len(rays), int(bool(shadow)) | (2 if direct else 0),
                                            hits.ctypes.data, sh.ctypes.data, cnt.ctypes.data, ev.ctypes.data, stride, ln.ctypes.data)
        if rc != 0:
            raise RuntimeError("orc_wide_trace_events failed: %d" % rc)
        return ev, ln

    def __del__(self):
        try:
            self.lib.orc_destroy(self.handle)
        except Exception:
            pass
