"""ctypes binding of the C-ABI in include/rt_hip.h (raytracing_amd/librt_hip.so).

Python here is plumbing for tests and bench.py; the product is the shared
library.  There is NO fallback: if the library is missing or no GPU is present
the constructors raise."""
import ctypes as C
import os
import numpy as np
from . import types as T

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librt_hip.so")
_lib = None


class RtError(RuntimeError):
    """Mirrors the reference's CLException (src/utils/cl_exception.hpp:109-123)."""


class rt_stats(C.Structure):
    _fields_ = [("closest_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("samples", C.c_uint64),
                ("last_active", C.c_uint32 * 64), ("last_shadow", C.c_uint32 * 64),
                ("samples_in_flight", C.c_uint32), ("samples_in_flight_limit", C.c_uint32), ("path_state_bytes", C.c_uint64),
                ("stack_spills", C.c_uint32), ("slow_rays", C.c_uint32), ("chunk_pixels", C.c_uint32), ("pipelines", C.c_uint32),
                ("log_inline_entries", C.c_uint32), ("log_fallbacks", C.c_uint32), ("frame_kernel_samples", C.c_uint32), ("samples_ahead", C.c_uint32), ("samples_from_banks", C.c_uint64)]


class rt_profile(C.Structure):
    _fields_ = [("ms_raygen", C.c_double), ("ms_trace_closest", C.c_double), ("ms_shade", C.c_double),
                ("ms_trace_shadow", C.c_double), ("n_raygen", C.c_uint32), ("n_trace_closest", C.c_uint32),
                ("n_shade", C.c_uint32), ("n_trace_shadow", C.c_uint32)]


class rt_scene_desc(C.Structure):
    _fields_ = [("triangles", C.c_void_p), ("num_triangles", C.c_uint32),
                ("nodes", C.c_void_p), ("num_nodes", C.c_uint32),
                ("materials", C.c_void_p), ("num_materials", C.c_uint32),
                ("textures", C.c_void_p), ("num_textures", C.c_uint32),
                ("texture_data", C.c_void_p), ("num_texture_data", C.c_uint32),
                ("lights", C.c_void_p), ("num_lights", C.c_uint32),
                ("emissive_indices", C.c_void_p), ("num_emissive", C.c_uint32),
                ("env_rgba", C.c_void_p), ("env_width", C.c_uint32), ("env_height", C.c_uint32),
                ("material_texture_indices", C.c_void_p), ("flags", C.c_uint32)]


SCENE_EMISSIVE_NEE = 1


class rt_frame_desc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("tile_rank", C.c_uint32),
                ("tile_count", C.c_uint32), ("band_height", C.c_uint32)]


EXPORTS = [
    "rt_ctx_create", "rt_ctx_destroy", "rt_finish", "rt_last_error", "rt_ctx_device_info", "rt_ctx_stream",
    "rt_ctx_set_option", "rt_upload_blue_noise_tables", "rt_host_register", "rt_host_unregister",
    "rt_buffer_create", "rt_buffer_destroy", "rt_buffer_write", "rt_buffer_read", "rt_buffer_copy",
    "rt_buffer_device_ptr", "rt_buffer_size", "rt_scene_upload", "rt_frame_create", "rt_frame_destroy",
    "rt_frame_local_rows", "rt_frame_global_row", "rt_set_option", "rt_set_camera", "rt_reset",
    "rt_generate_rays", "rt_intersect", "rt_shade_miss", "rt_clear_outgoing_counter", "rt_clear_shadow_counter",
    "rt_shade", "rt_intersect_shadow", "rt_accumulate_direct", "rt_advance_sample", "rt_integrate",
    "rt_frame_reserve_samples",
    "rt_compute_aovs", "rt_denoise", "rt_copy_history",
    "rt_frame_resolve", "rt_frame_present", "rt_frame_present_wait", "rt_frame_read_radiance", "rt_frame_radiance_device_ptr", "rt_frame_sample_count",
    "rt_frame_get_stats", "rt_frame_get_profile", "rt_frame_copy_radiance", "rt_frame_debug_read_queue", "rt_frame_debug_read_hits", "rt_debug_eval",
    "rt_debug_wide_bvh", "rt_frame_debug_timeline", "rt_frame_debug_frame_rows", "rt_debug_own_bvh", "rt_debug_wide_bvh_metric", "rt_scene_tree_report", "rt_debug_choose_tree", "rt_debug_adapt_fold", "rt_debug_fold_abandon", "rt_debug_adapt_shadow_side", "rt_debug_rotate_tree", "rt_debug_fold_view_left", "rt_debug_device_fold", "rt_debug_wide_bvh_weights", "rt_debug_pair_layout", "rt_debug_device_tree", "rt_scene_export_folds", "rt_scene_import_folds",
    "rt_group_create", "rt_group_unique_id", "rt_group_join", "rt_group_size", "rt_group_local_count", "rt_group_local_rank", "rt_group_comm_count",
    "rt_group_gather_radiance", "rt_group_destroy", "rt_group_last_error", "rt_group_denoise", "rt_group_create_local",
    "rt_group_create_unchecked",
]

OPT_MAX_BOUNCES, OPT_WHITE_FURNACE, OPT_SAMPLER, OPT_AOV, OPT_DENOISER, OPT_DROP_LAST, OPT_PROFILE, OPT_TRACE_VARIANT, OPT_TRACE_WAVES, OPT_SAMPLES_IN_FLIGHT, OPT_SELECT_FORM_BOX, OPT_PACKET_BOUNCES, OPT_TRACE_TUNE, OPT_DEBUG_ALLOC_LIMIT, OPT_PATH_STATE_LIMIT_MB, OPT_PIPELINES, OPT_SHADE_PARTITION, OPT_OVERLAP_SHADOW, OPT_SMALL_LAUNCH_PATHS, OPT_COMPACT_LOG, OPT_DEBUG_LOG_POOL_DIV, OPT_TRACE_TAIL_LANES, OPT_TRACE_TAIL_PATHS, OPT_CHUNK_REFILL, OPT_STAGE_PIPES, OPT_FRAME_KERNEL, OPT_SAMPLES_AHEAD = range(27)


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtError("librt_hip.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`)")
    lib = C.CDLL(LIB_PATH)
    vp, u32, i32, sz = C.c_void_p, C.c_uint32, C.c_int, C.c_size_t
    sig = {
        "rt_ctx_create": (i32, [i32, C.POINTER(vp)]), "rt_ctx_destroy": (i32, [vp]), "rt_finish": (i32, [vp]),
        "rt_last_error": (C.c_char_p, [vp]),
        "rt_ctx_device_info": (i32, [vp, C.c_char_p, sz, C.POINTER(i32), C.POINTER(sz)]),
        "rt_ctx_stream": (vp, [vp]), "rt_ctx_set_option": (i32, [vp, i32, u32]),
        "rt_upload_blue_noise_tables": (i32, [vp, vp, vp, vp]),
        "rt_host_register": (i32, [vp, vp, sz]), "rt_host_unregister": (i32, [vp, vp]),
        "rt_buffer_create": (i32, [vp, sz, vp, C.POINTER(vp)]), "rt_buffer_destroy": (i32, [vp]),
        "rt_buffer_write": (i32, [vp, sz, vp, sz]), "rt_buffer_read": (i32, [vp, sz, vp, sz]),
        "rt_buffer_copy": (i32, [vp, vp, sz, sz, sz]), "rt_buffer_device_ptr": (vp, [vp]),
        "rt_buffer_size": (sz, [vp]),
        "rt_scene_upload": (i32, [vp, C.POINTER(rt_scene_desc)]),
        "rt_frame_create": (i32, [vp, C.POINTER(rt_frame_desc), C.POINTER(vp)]), "rt_frame_destroy": (i32, [vp]),
        "rt_frame_local_rows": (u32, [vp]), "rt_frame_global_row": (u32, [vp, u32]),
        "rt_set_option": (i32, [vp, i32, u32]), "rt_set_camera": (i32, [vp, vp]),
        "rt_reset": (i32, [vp]), "rt_generate_rays": (i32, [vp]), "rt_intersect": (i32, [vp, u32]),
        "rt_shade_miss": (i32, [vp, u32]), "rt_clear_outgoing_counter": (i32, [vp, u32]),
        "rt_clear_shadow_counter": (i32, [vp]), "rt_shade": (i32, [vp, u32]),
        "rt_intersect_shadow": (i32, [vp, u32]), "rt_accumulate_direct": (i32, [vp]),
        "rt_advance_sample": (i32, [vp]), "rt_integrate": (i32, [vp, u32]),
        "rt_frame_reserve_samples": (i32, [vp, u32, C.POINTER(C.c_uint32)]),
        "rt_compute_aovs": (i32, [vp]), "rt_denoise": (i32, [vp]), "rt_copy_history": (i32, [vp]),
        "rt_frame_resolve": (i32, [vp, vp]), "rt_frame_present": (i32, [vp, vp]), "rt_frame_present_wait": (i32, [vp]),
        "rt_frame_read_radiance": (i32, [vp, vp]),
        "rt_frame_radiance_device_ptr": (vp, [vp]), "rt_frame_sample_count": (u32, [vp]),
        "rt_frame_get_stats": (i32, [vp, C.POINTER(rt_stats)]),
        "rt_frame_get_profile": (i32, [vp, C.POINTER(rt_profile)]),
        "rt_frame_copy_radiance": (i32, [vp, vp]),
        "rt_frame_debug_read_queue": (i32, [vp, i32, u32, vp, vp, vp, u32, C.POINTER(u32)]),
        "rt_frame_debug_read_hits": (i32, [vp, vp, u32]),
        "rt_debug_eval": (i32, [vp, i32, vp, vp, vp, u32]),
        "rt_debug_wide_bvh": (i32, [vp, u32, i32, vp, vp, u32, C.POINTER(u32), C.POINTER(u32)]),
        "rt_frame_debug_timeline": (i32, [vp, i32, vp]),
        "rt_frame_debug_frame_rows": (i32, [vp, vp, u32, C.POINTER(u32), C.POINTER(u32)]),
        "rt_scene_tree_report": (C.c_char_p, [vp]),
        "rt_debug_choose_tree": (i32, [C.POINTER(rt_scene_desc), i32, u32, vp, u32, C.POINTER(u32), C.POINTER(u32), C.c_char_p, sz]),
        "rt_debug_own_bvh": (i32, [vp, u32, C.c_double, vp, u32, vp, u32, C.POINTER(u32)]),
        "rt_debug_wide_bvh_metric": (i32, [vp, u32, C.c_double, vp, u32, vp, u32, C.POINTER(u32), C.POINTER(u32)]),
        "rt_debug_fold_view_left": (i32, [vp, vp, C.c_double]),
        "rt_debug_device_fold": (i32, [vp, vp, u32, C.c_double, vp, u32, vp, vp, vp, u32, C.POINTER(u32), C.POINTER(u32), C.POINTER(C.c_double)]),
        "rt_debug_wide_bvh_weights": (i32, [vp, u32, vp, vp, vp, u32, C.POINTER(u32), C.POINTER(u32)]),
        "rt_debug_pair_layout": (i32, [vp, u32, vp, vp, u32]),
        "rt_debug_device_tree": (i32, [vp, vp, u32, C.c_double, vp, u32, vp, u32, C.POINTER(u32), C.POINTER(C.c_double), C.POINTER(u32), u32, vp, C.c_double]),
        "rt_scene_export_folds": (i32, [vp, vp, vp, u32, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]),
        "rt_scene_import_folds": (i32, [vp, vp, u32, u32, vp, u32, u32]),
        "rt_debug_rotate_tree": (i32, [vp, u32, vp, vp, u32, i32, vp, C.POINTER(C.c_double), C.POINTER(u32), i32, C.c_double]),
        "rt_debug_adapt_shadow_side": (i32, [vp, u32, vp, vp, u32, u32, vp, vp, u32, C.POINTER(u32), C.POINTER(u32), vp, C.POINTER(C.c_double), C.POINTER(u32), vp, u32, C.POINTER(u32)]),
        "rt_debug_fold_abandon": (C.c_double, [vp, u32, vp, vp, u32, u32, u32, C.POINTER(i32)]),
        "rt_debug_adapt_fold": (i32, [vp, u32, vp, vp, u32, vp, vp, u32, C.POINTER(u32), C.POINTER(u32), C.POINTER(C.c_double), C.POINTER(i32)]),
        "rt_group_create": (i32, [i32, C.POINTER(i32), C.POINTER(vp)]), "rt_group_create_unchecked": (i32, [i32, C.POINTER(i32), C.POINTER(vp)]),
        "rt_group_unique_id": (i32, [vp, sz]),
        "rt_group_join": (i32, [i32, i32, vp, i32, C.POINTER(vp)]), "rt_group_size": (i32, [vp]),
        "rt_group_local_count": (i32, [vp]), "rt_group_local_rank": (i32, [vp, i32]),
        "rt_group_comm_count": (i32, [vp, i32, C.POINTER(i32), C.POINTER(i32)]),
        "rt_group_gather_radiance": (i32, [vp, C.POINTER(vp), i32, vp, C.POINTER(vp)]), "rt_group_destroy": (i32, [vp]),
        "rt_group_last_error": (C.c_char_p, [vp]),
        "rt_group_denoise": (i32, [vp, C.POINTER(vp), i32, vp, vp]), "rt_group_create_local": (i32, [i32, i32, C.POINTER(vp)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def _check(lib, ctx, rc):
    if rc != 0:
        msg = lib.rt_last_error(ctx)
        raise RtError(msg.decode() if msg else "unknown error")


def choose_tree(scene, shadow=True, mode=1):
    """rt_debug_choose_tree (host only, no GPU): the 4-wide tree rt_scene_upload would give this scene's shadow / closest-hit
    rays under RT_CTX_OPT_SHADOW_TREE / RT_CTX_OPT_CLOSEST_TREE = mode.  scene: dict with triangles, nodes, lights.
    Returns (records as bytes-compatible uint8[n, 64], entry_ref, report)."""
    lib = load()
    tris, nodes, lights = (np.ascontiguousarray(scene[k]) for k in ("triangles", "nodes", "lights"))
    d = rt_scene_desc()
    d.triangles, d.num_triangles, d.nodes, d.num_nodes = tris.ctypes.data, len(tris), nodes.ctypes.data, len(nodes)
    d.lights, d.num_lights = (lights.ctypes.data if len(lights) else None), len(lights)
    n, entry = C.c_uint32(), C.c_uint32()
    rep = C.create_string_buffer(2048)
    if lib.rt_debug_choose_tree(C.byref(d), int(bool(shadow)), mode, None, 0, C.byref(n), C.byref(entry), rep, len(rep)):
        raise RtError(lib.rt_last_error(None).decode())
    out = np.zeros((n.value, 64), np.uint8)
    if lib.rt_debug_choose_tree(C.byref(d), int(bool(shadow)), mode, out.ctypes.data, n.value, C.byref(n), C.byref(entry), rep, len(rep)):
        raise RtError(lib.rt_last_error(None).decode())
    return out, entry.value, rep.value.decode()


ADAPTIVE_FOLD_DEFAULT = 25     # rt_ctx's RT_CTX_OPT_ADAPTIVE_FOLD as created (rt_hip.hip): bits 0 + 3 + 4 since round 5


def export_folds(ctx_handle):
    """rt_scene_export_folds: (closest records uint8[n, 64], shadow records uint8[m, 64] (m = 0: shared), (closest entry, shadow entry))"""
    lib = load()
    n, m = C.c_uint32(), C.c_uint32()
    ent = (C.c_uint32 * 2)()
    if lib.rt_scene_export_folds(ctx_handle, None, None, 0, C.byref(n), C.byref(m), ent):
        raise RtError(lib.rt_last_error(ctx_handle).decode())
    cap = max(n.value, m.value, 1)
    cl, sh = np.zeros((cap, 64), np.uint8), np.zeros((cap, 64), np.uint8)
    if lib.rt_scene_export_folds(ctx_handle, cl.ctypes.data, sh.ctypes.data, cap, C.byref(n), C.byref(m), ent):
        raise RtError(lib.rt_last_error(ctx_handle).decode())
    return cl[:n.value].copy(), sh[:m.value].copy(), (int(ent[0]), int(ent[1]))


def import_folds(ctx_handle, closest, shadow, entries):
    """rt_scene_import_folds: another context's records in place of this one's (same scene); its own adaptation goes off"""
    lib = load()
    cl = np.ascontiguousarray(closest, np.uint8)
    sh = np.ascontiguousarray(shadow, np.uint8) if shadow is not None and len(shadow) else None
    if lib.rt_scene_import_folds(ctx_handle, cl.ctypes.data, len(cl), entries[0], sh.ctypes.data if sh is not None else None, len(sh) if sh is not None else 0, entries[1]):
        raise RtError(lib.rt_last_error(ctx_handle).decode())


def device_fold(ctx, nodes, iso_weight=-1.0, dirs=None, weights=None):
    """rt_debug_device_fold: the SAH collapse of the LinearBVHNode[] `nodes` on ctx's device (raytracing_amd/csrc/fold_kernels.h).
    iso_weight < 0: the plain surface area; else the metric of rt_debug_wide_bvh_metric; weights: float64[n] per node or None.
    Returns (records uint8[n, 64], entry_ref, roots uint32[n], seconds)."""
    lib = load()
    nodes = np.ascontiguousarray(nodes)
    d = np.ascontiguousarray(dirs, np.float32) if dirs is not None else None
    w = np.ascontiguousarray(weights, np.float64) if weights is not None else None
    cap = len(nodes) // 2 + 2
    out, roots = np.zeros((cap, 64), np.uint8), np.zeros(cap, np.uint32)
    n, entry, sec = C.c_uint32(), C.c_uint32(), C.c_double()
    if lib.rt_debug_device_fold(ctx.handle, nodes.ctypes.data, len(nodes), iso_weight, d.ctypes.data if d is not None else None, len(d) if d is not None else 0,
                                w.ctypes.data if w is not None else None, out.ctypes.data, roots.ctypes.data, cap, C.byref(n), C.byref(entry), C.byref(sec)):
        raise RtError(lib.rt_last_error(ctx.handle).decode())
    return out[:n.value].copy(), entry.value, roots[:n.value].copy(), sec.value


def device_tree(ctx, nodes, iso_weight=1.0, dirs=None, radius=0, frame_dir=None, stretch=1.0):
    """rt_debug_device_tree: a binary tree over the leaves of `nodes` built on ctx's device (PLOC, raytracing_amd/csrc/ploc_kernels.h).  Returns (nodes, seconds, rounds)."""
    lib = load()
    nodes = np.ascontiguousarray(nodes)
    d = np.ascontiguousarray(np.asarray(dirs, np.float32).reshape(-1, 3)) if dirs is not None else None
    out = np.zeros(len(nodes), nodes.dtype)
    n, sec, rounds = C.c_uint32(), C.c_double(), C.c_uint32()
    if lib.rt_debug_device_tree(ctx.handle, nodes.ctypes.data, len(nodes), iso_weight, d.ctypes.data if d is not None and len(d) else None, len(d) if d is not None else 0,
                                out.ctypes.data, len(out), C.byref(n), C.byref(sec), C.byref(rounds), radius,
                                np.ascontiguousarray(frame_dir, np.float32).ctypes.data if frame_dir is not None else None, stretch):
        raise RtError(lib.rt_last_error(ctx.handle).decode())
    return out[:n.value].copy(), sec.value, rounds.value


def wide_bvh_weights(nodes, weights):
    """rt_debug_wide_bvh_weights (host only): build_wide_bvh's SAH collapse with per-node weights in place of the area.  Returns (records, entry_ref, roots)."""
    lib = load()
    nodes = np.ascontiguousarray(nodes)
    w = np.ascontiguousarray(weights, np.float64)
    cap = len(nodes) // 2 + 2
    out, roots = np.zeros((cap, 64), np.uint8), np.zeros(cap, np.uint32)
    n, entry = C.c_uint32(), C.c_uint32()
    if lib.rt_debug_wide_bvh_weights(nodes.ctypes.data, len(nodes), w.ctypes.data, out.ctypes.data, roots.ctypes.data, cap, C.byref(n), C.byref(entry)):
        raise RtError(lib.rt_last_error(None).decode())
    return out[:n.value].copy(), entry.value, roots[:n.value].copy()


def adapt_fold(nodes, origins_tmax, directions):
    """rt_debug_adapt_fold (host only, no GPU): RT_CTX_OPT_ADAPTIVE_FOLD's re-fold of the LinearBVHNode[] `nodes` for the rays given
    (origins_tmax float32[n, 4] = x, y, z, t_max; directions float32[n, 4] = x, y, z, -).
    Returns (records uint8[n, 64], entry_ref, roots uint32[n], (cost of the surface-area fold, cost of the adapted fold), adopted)."""
    lib = load()
    nodes = np.ascontiguousarray(nodes)
    o = np.ascontiguousarray(origins_tmax, np.float32).reshape(-1, 4)
    d = np.ascontiguousarray(directions, np.float32).reshape(-1, 4)
    assert len(o) == len(d)
    n, entry, cheaper = C.c_uint32(), C.c_uint32(), C.c_int()
    cost = (C.c_double * 2)()
    out = np.zeros((len(nodes), 64), np.uint8)                  # a fold never has more records than the tree has nodes
    roots = np.zeros(len(nodes), np.uint32)
    if lib.rt_debug_adapt_fold(nodes.ctypes.data, len(nodes), o.ctypes.data, d.ctypes.data, len(o), out.ctypes.data, roots.ctypes.data, len(out), C.byref(n), C.byref(entry),
                               cost, C.byref(cheaper)):
        raise RtError(lib.rt_last_error(None).decode())
    return out[:n.value].copy(), entry.value, roots[:n.value].copy(), (cost[0], cost[1]), bool(cheaper.value)


def adapt_shadow_side(nodes, origins_tmax, directions, mode, triangles=None):
    """rt_debug_adapt_shadow_side (host only): what FoldAdapt's worker does for the shadow rays under RT_CTX_OPT_ADAPTIVE_FOLD = mode
    (triangles: the scene's rt_triangle array, needed for bit 4).
    Returns (records uint8[n, 64], entry_ref, roots, the tree the records fold, (current cost, candidate cost), rotations, adopted, records reordered)."""
    lib = load()
    nodes = np.ascontiguousarray(nodes)
    o = np.ascontiguousarray(origins_tmax, np.float32).reshape(-1, 4)
    d = np.ascontiguousarray(directions, np.float32).reshape(-1, 4)
    tris = np.ascontiguousarray(triangles) if triangles is not None else None
    n, entry, made, moved = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    cost = (C.c_double * 2)()
    out = np.zeros((len(nodes), 64), np.uint8)
    roots = np.zeros(len(nodes), np.uint32)
    tree = np.zeros(len(nodes), nodes.dtype)
    rc = lib.rt_debug_adapt_shadow_side(nodes.ctypes.data, len(nodes), o.ctypes.data, d.ctypes.data, len(o), mode, out.ctypes.data, roots.ctypes.data, len(out),
                                        C.byref(n), C.byref(entry), tree.ctypes.data, cost, C.byref(made),
                                        tris.ctypes.data if tris is not None else None, len(tris) if tris is not None else 0, C.byref(moved))
    if rc < 0:
        raise RtError(lib.rt_last_error(None).decode())
    return out[:n.value].copy(), entry.value, roots[:n.value].copy(), tree, (cost[0], cost[1]), made.value, bool(rc), moved.value


def rotate_tree(nodes, origins_tmax, directions, max_passes=8, moves=3, min_gain=0.03):
    """rt_debug_rotate_tree (host only): tree_rotate.h's local search on the binary tree `nodes` for the rays given.
    Returns (rotated nodes, (crossings per ray before, after), rotations made)."""
    lib = load()
    nodes = np.ascontiguousarray(nodes)
    o = np.ascontiguousarray(origins_tmax, np.float32).reshape(-1, 4)
    d = np.ascontiguousarray(directions, np.float32).reshape(-1, 4)
    out = np.zeros(len(nodes), nodes.dtype)
    cost = (C.c_double * 2)()
    made = C.c_uint32()
    if lib.rt_debug_rotate_tree(nodes.ctypes.data, len(nodes), o.ctypes.data, d.ctypes.data, len(o), max_passes, out.ctypes.data, cost, C.byref(made), moves, min_gain):
        raise RtError(lib.rt_last_error(None).decode())
    return out, (cost[0], cost[1]), made.value


class Context:
    """CLContext replacement (src/gpu_wrappers/cl_context.hpp:37-65)."""

    def __init__(self, device=0):
        self.lib = load()
        h = C.c_void_p()
        _check(self.lib, None, self.lib.rt_ctx_create(device, C.byref(h)))
        self.handle = h
        self._frames = []          # weakrefs: frames must be destroyed before their context

    def device_info(self):
        name = C.create_string_buffer(256)
        cu = C.c_int()
        mem = C.c_size_t()
        _check(self.lib, self.handle, self.lib.rt_ctx_device_info(self.handle, name, 256, C.byref(cu), C.byref(mem)))
        return name.value.decode(), cu.value, mem.value

    def stream(self):
        return self.lib.rt_ctx_stream(self.handle)

    def upload_blue_noise_tables(self, sobol, scrambling, ranking):
        t = [np.ascontiguousarray(x, np.int32) for x in (sobol, scrambling, ranking)]
        _check(self.lib, self.handle, self.lib.rt_upload_blue_noise_tables(self.handle, *[x.ctypes.data for x in t]))

    def set_treelet_nodes(self, n):
        _check(self.lib, self.handle, self.lib.rt_ctx_set_option(self.handle, 0, n))

    def set_wide_bvh(self, mode):
        """RT_CTX_OPT_WIDE_BVH: 1 = SAH-optimal frontier per wide record (default), 2 = two BVH2 levels per record, 0 = none"""
        _check(self.lib, self.handle, self.lib.rt_ctx_set_option(self.handle, 1, mode))

    def set_shadow_tree(self, mode):
        """RT_CTX_OPT_SHADOW_TREE (effective at the next upload_scene): 1 default (own tree where it measures cheaper), 2 own always,
        3 own with the surface-area metric, 0 shared with the closest-hit rays.  Results are bit-identical for every value."""
        _check(self.lib, self.handle, self.lib.rt_ctx_set_option(self.handle, 2, mode))

    def set_closest_tree(self, mode):
        """RT_CTX_OPT_CLOSEST_TREE: 0 default (bit-identical), 1 / 2 = tolerance mode (own tree where cheaper / always)"""
        _check(self.lib, self.handle, self.lib.rt_ctx_set_option(self.handle, 3, mode))

    def set_adaptive_fold(self, mode):
        """RT_CTX_OPT_ADAPTIVE_FOLD (effective at the next upload_scene): bit 0 = the first integrate() probes the frame's own rays and the
        4-wide trees are folded again for them (exact), bit 1 = integrate() waits for the new fold, bit 2 = small trees too."""
        _check(self.lib, self.handle, self.lib.rt_ctx_set_option(self.handle, 4, mode))

    def set_adapt_min_interval_ms(self, ms):
        """RT_CTX_OPT_ADAPT_MIN_INTERVAL_MS (default 500): a camera that keeps leaving the adapted view starts at most one fold
        adaptation per this many milliseconds (not applied when bit 1 of the adaptive-fold mode waits for every adaptation)"""
        _check(self.lib, self.handle, self.lib.rt_ctx_set_option(self.handle, 5, ms))

    def tree_report(self):
        return self.lib.rt_scene_tree_report(self.handle).decode()

    def finish(self):
        _check(self.lib, self.handle, self.lib.rt_finish(self.handle))

    def upload_scene(self, scene):
        """scene: dict with triangles (BVH order), nodes, materials, textures,
        texture_data, lights, emissive, env (H x W x 4 float32)."""
        s = {k: np.ascontiguousarray(v) for k, v in scene.items() if k not in ("scene_info", "flags")}
        for key, dt in (("triangles", T.triangle), ("nodes", T.bvh_node), ("materials", T.packed_material),
                        ("textures", T.texture), ("lights", T.light)):
            if s[key].dtype != dt:
                raise RtError("scene['%s'] has the wrong dtype" % key)
        env = s["env"].astype(np.float32, copy=False)
        p = lambda a: a.ctypes.data if a.size else None
        d = rt_scene_desc(p(s["triangles"]), len(s["triangles"]), p(s["nodes"]), len(s["nodes"]),
                          p(s["materials"]), len(s["materials"]), p(s["textures"]), len(s["textures"]),
                          p(s["texture_data"]), len(s["texture_data"]), p(s["lights"]), len(s["lights"]),
                          p(s["emissive"]), len(s["emissive"]), p(env), env.shape[1], env.shape[0])
        # opt-in extensions: 6 x uint16 texture indices per material, RT_SCENE_* flags
        tex16 = s.get("material_texture_indices")
        if tex16 is not None:
            tex16 = np.ascontiguousarray(tex16, np.uint16)
            if tex16.size != 6 * len(s["materials"]):
                raise RtError("scene['material_texture_indices'] needs 6 entries per material")
            d.material_texture_indices = tex16.ctypes.data
        d.flags = int(scene.get("flags", 0))
        _check(self.lib, self.handle, self.lib.rt_scene_upload(self.handle, C.byref(d)))

    def debug_eval(self, fn, a, b=None):
        a = np.ascontiguousarray(a, np.float32)
        out = np.zeros_like(a)
        bp = None
        if b is not None:
            b = np.ascontiguousarray(b, np.float32)
            bp = b.ctypes.data
        _check(self.lib, self.handle, self.lib.rt_debug_eval(self.handle, fn, a.ctypes.data, bp, out.ctypes.data,
                                                             a.size))
        return out

    def close(self):
        if self.handle:
            for ref in self._frames:
                fr = ref()
                if fr is not None:
                    fr.close()
            self._frames = []
            self.lib.rt_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Frame:
    """Device half of CLPathTraceIntegrator for one tile of the image."""

    def __init__(self, ctx, width, height, tile_rank=0, tile_count=1, band_height=8):
        self.ctx, self.lib = ctx, ctx.lib
        self.width, self.height = width, height
        d = rt_frame_desc(width, height, tile_rank, tile_count, band_height)
        h = C.c_void_p()
        _check(self.lib, ctx.handle, self.lib.rt_frame_create(ctx.handle, C.byref(d), C.byref(h)))
        self.handle = h
        self.local_rows = self.lib.rt_frame_local_rows(h)
        import weakref
        ctx._frames.append(weakref.ref(self))

    def _c(self, rc):
        _check(self.lib, self.ctx.handle, rc)

    def global_rows(self):
        return np.array([self.lib.rt_frame_global_row(self.handle, r) for r in range(self.local_rows)], np.int64)

    def set_camera(self, cam):
        cam = np.ascontiguousarray(cam)
        self._c(self.lib.rt_set_camera(self.handle, cam.ctypes.data))

    def set_option(self, opt, value):
        self._c(self.lib.rt_set_option(self.handle, opt, value))

    def set_max_bounces(self, b):
        self.set_option(OPT_MAX_BOUNCES, b)

    def reset(self):
        self._c(self.lib.rt_reset(self.handle))

    def integrate(self, n=1):
        self._c(self.lib.rt_integrate(self.handle, n))

    def reserve_samples(self, n):
        got = C.c_uint32(0)
        self._c(self.lib.rt_frame_reserve_samples(self.handle, n, C.byref(got)))
        return got.value

    # stage API
    def generate_rays(self): self._c(self.lib.rt_generate_rays(self.handle))
    def intersect(self, b): self._c(self.lib.rt_intersect(self.handle, b))
    def shade(self, b): self._c(self.lib.rt_shade(self.handle, b))
    def intersect_shadow(self, b): self._c(self.lib.rt_intersect_shadow(self.handle, b))
    def advance_sample(self): self._c(self.lib.rt_advance_sample(self.handle))

    def radiance(self):
        out = np.zeros((self.local_rows, self.width, 4), np.float32)
        self._c(self.lib.rt_frame_read_radiance(self.handle, out.ctypes.data))
        return out

    def resolve(self):
        out = np.zeros((self.local_rows, self.width, 4), np.float32)
        self._c(self.lib.rt_frame_resolve(self.handle, out.ctypes.data))
        return out

    def present(self, out=None):
        """rt_frame_present: resolve + Finish() on the frame's kernels; the image travels to `out` (kept alive by the caller) on a
        copy stream and is complete after present_wait().  Returns `out`."""
        if out is None:
            out = np.zeros((self.local_rows, self.width, 4), np.float32)
        self._c(self.lib.rt_frame_present(self.handle, out.ctypes.data))
        return out

    def present_wait(self):
        self._c(self.lib.rt_frame_present_wait(self.handle))

    def radiance_device_ptr(self):
        return self.lib.rt_frame_radiance_device_ptr(self.handle)

    def sample_count(self):
        return self.lib.rt_frame_sample_count(self.handle)

    def stats(self):
        st = rt_stats()
        self._c(self.lib.rt_frame_get_stats(self.handle, C.byref(st)))
        return st

    def profile(self):
        p = rt_profile()
        self._c(self.lib.rt_frame_get_profile(self.handle, C.byref(p)))
        return p

    def copy_radiance_to(self, device_ptr):
        self._c(self.lib.rt_frame_copy_radiance(self.handle, device_ptr))

    def read_queue(self, which, bounce):
        cnt = C.c_uint32()
        # two-call pattern: size query first (the queue holds up to samples-in-flight x tile pixels entries)
        self._c(self.lib.rt_frame_debug_read_queue(self.handle, which, bounce, None, None, None, 0, C.byref(cnt)))
        n = cnt.value
        rays = np.zeros(max(n, 1), T.ray)
        pix = np.zeros(max(n, 1), np.uint32)
        payload = np.zeros(max(n, 1), T.float4)
        self._c(self.lib.rt_frame_debug_read_queue(self.handle, which, bounce, rays.ctypes.data, pix.ctypes.data,
                                                   payload.ctypes.data, n, C.byref(cnt)))
        assert cnt.value == n
        return rays[:n], pix[:n], payload[:n]

    def read_hits(self, count):
        hits = np.zeros(count, T.hit)
        self._c(self.lib.rt_frame_debug_read_hits(self.handle, hits.ctypes.data, count))
        return hits

    def close(self):
        if self.handle:
            self.lib.rt_frame_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Group:
    """rt_group: ranks of one tiled image and their single RCCL gather (include/rt_hip.h, "device groups").

      Group.create([0, 1, ...])            all ranks in this process (one per device)
      Group.join(nranks, rank, id, device) one process per GPU; id = Group.unique_id() made by rank 0 and
                                           carried to the others by the launcher's own channel"""

    ID_BYTES = 128

    def __init__(self, handle):
        self.lib = load()
        self.handle = handle

    @classmethod
    def create(cls, devices, unchecked=False):
        """unchecked: rt_group_create_unchecked -- the device list goes to ncclCommInitAll as it is (wiring test)"""
        lib = load()
        arr = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        if (lib.rt_group_create_unchecked if unchecked else lib.rt_group_create)(len(devices), arr, C.byref(h)) != 0:
            raise RtError(lib.rt_group_last_error(None).decode())
        return cls(h)

    @classmethod
    def create_local(cls, n, device=0):
        """n ranks on ONE device, device copies instead of RCCL: the test / plumbing transport."""
        lib = load()
        h = C.c_void_p()
        if lib.rt_group_create_local(n, device, C.byref(h)) != 0:
            raise RtError(lib.rt_group_last_error(None).decode())
        return cls(h)

    @staticmethod
    def unique_id():
        lib = load()
        buf = C.create_string_buffer(Group.ID_BYTES)
        if lib.rt_group_unique_id(buf, Group.ID_BYTES) != 0:
            raise RtError(lib.rt_group_last_error(None).decode())
        return buf.raw

    @classmethod
    def join(cls, nranks, rank, id_bytes, device):
        lib = load()
        assert len(id_bytes) == Group.ID_BYTES
        h = C.c_void_p()
        if lib.rt_group_join(nranks, rank, id_bytes, device, C.byref(h)) != 0:
            raise RtError(lib.rt_group_last_error(None).decode())
        return cls(h)

    def size(self):
        return self.lib.rt_group_size(self.handle)

    def local_ranks(self):
        return [self.lib.rt_group_local_rank(self.handle, i) for i in range(self.lib.rt_group_local_count(self.handle))]

    def comm_count(self, i=0):
        """(ncclCommCount, ncclCommUserRank) of local member i's communicator -- RCCL's own word on how many ranks the
        gather spans; (0, -1) for a local group (device copies, no RCCL)."""
        n, r = C.c_int(0), C.c_int(-1)
        if self.lib.rt_group_comm_count(self.handle, i, C.byref(n), C.byref(r)) != 0:
            raise RtError(self.lib.rt_group_last_error(self.handle).decode())
        return n.value, r.value

    def gather_radiance(self, frame_handles, root, height, width, want_host=True):
        """frame_handles: rt_frame* of the local members, in member order.  Returns the image (numpy,
        height x width x 4) on the process that owns `root` when want_host, else None."""
        n = len(frame_handles)
        arr = (C.c_void_p * n)(*[h if isinstance(h, int) else h.value for h in frame_handles])
        owns_root = root in self.local_ranks()
        out = np.zeros((height, width, 4), np.float32) if (owns_root and want_host) else None
        dev = C.c_void_p()
        rc = self.lib.rt_group_gather_radiance(self.handle, arr, root, out.ctypes.data if out is not None else None, C.byref(dev))
        if rc != 0:
            raise RtError(self.lib.rt_group_last_error(self.handle).decode())
        self.device_image = dev.value
        return out

    def denoise(self, frame_handles, root, height, width):
        """Gather-then-denoise on the root (rt_group_denoise).  Returns (resolved, radiance) on the root's process."""
        n = len(frame_handles)
        arr = (C.c_void_p * n)(*[h if isinstance(h, int) else h.value for h in frame_handles])
        owns_root = root in self.local_ranks()
        res = np.zeros((height, width, 4), np.float32) if owns_root else None
        rad = np.zeros((height, width, 4), np.float32) if owns_root else None
        rc = self.lib.rt_group_denoise(self.handle, arr, root, res.ctypes.data if owns_root else None, rad.ctypes.data if owns_root else None)
        if rc != 0:
            raise RtError(self.lib.rt_group_last_error(self.handle).decode())
        return res, rad

    def close(self):
        if self.handle:
            self.lib.rt_group_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
