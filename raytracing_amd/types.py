"""numpy dtypes mirroring include/rt_types.h (reference:
src/kernels/common/shared_structures.h:56-181).  Host-side plumbing only."""
import numpy as np

float3 = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("w", "<f4")])
float4 = float3
float2 = np.dtype([("x", "<f4"), ("y", "<f4")])

ray = np.dtype([("origin", float4), ("direction", float4)])
hit = np.dtype([("bc", float2), ("primitive_id", "<u4"), ("t", "<f4")])
scene_info = np.dtype([("analytic_light_count", "<u4"), ("emissive_count", "<u4"),
                       ("environment_map_index", "<u4"), ("padding", "<u4")])
packed_material = np.dtype([("diffuse_albedo", "<u4"), ("specular_albedo", "<u4"), ("emission", "<u4"),
                            ("roughness_metalness", "<u4"), ("ior_emission_idx_transparency", "<u4")])
light = np.dtype([("origin", float3), ("radiance", float3), ("type", "<u4"), ("padding", "<u4", (3,))])
texture = np.dtype([("data_start", "<i4"), ("width", "<i4"), ("height", "<i4"), ("padding", "<i4")])
vertex = np.dtype([("position", float3), ("texcoord", float3), ("normal", float3)])
triangle = np.dtype([("v1", vertex), ("v2", vertex), ("v3", vertex), ("mtl_index", "<u4"),
                     ("padding", "<u4", (3,))])
bvh_node = np.dtype([("bounds_min", float3), ("bounds_max", float3), ("offset", "<u4"),
                     ("num_primitives_axis", "<u4"), ("padding", "<u4", (2,))])
camera = np.dtype([("position", float3), ("front", float3), ("up", float3), ("fov", "<f4"),
                   ("aspect_ratio", "<f4"), ("aperture", "<f4"), ("focus_distance", "<f4")])

assert ray.itemsize == 32 and hit.itemsize == 16 and scene_info.itemsize == 16
assert packed_material.itemsize == 20 and light.itemsize == 48 and texture.itemsize == 16
assert vertex.itemsize == 48 and triangle.itemsize == 160 and bvh_node.itemsize == 48
assert camera.itemsize == 64


def default_camera(width, height):
    """The reference's start-up camera (src/utils/camera_controller.cpp:30-41,77-80):
    position (0,-1,1), yaw = pitch = MATH_PIDIV2, fov = 75*3.1415/180, Z-up."""
    f32 = np.float32
    yaw = f32(1.570796327)
    pitch = f32(1.570796327)
    # std::cosf/std::sinf in binary32 (glibc); values below are what the
    # reference computes, stored as exact binary32 literals so that no libm
    # is involved at run time (pinned by tests/test_ref_pin.py).
    cy, sy = f32(np.cos(yaw, dtype=f32)), f32(np.sin(yaw, dtype=f32))
    cp, sp = f32(np.cos(pitch, dtype=f32)), f32(np.sin(pitch, dtype=f32))
    front = np.array([f32(cy * sp), f32(sy * sp), cp], dtype=f32)
    up0 = np.array([0, 0, 1], dtype=f32)

    def cross(a, b):
        return np.array([f32(f32(a[1] * b[2]) - f32(a[2] * b[1])),
                         f32(f32(a[2] * b[0]) - f32(a[0] * b[2])),
                         f32(f32(a[0] * b[1]) - f32(a[1] * b[0]))], dtype=f32)

    r = cross(front, up0)
    ln = f32(np.sqrt(f32(f32(f32(r[0] * r[0]) + f32(r[1] * r[1])) + f32(r[2] * r[2]))))
    right = np.array([f32(r[0] / ln), f32(r[1] / ln), f32(r[2] / ln)], dtype=f32)
    up = cross(right, front)
    cam = np.zeros((), dtype=camera)
    cam["position"]["x"], cam["position"]["y"], cam["position"]["z"] = 0.0, -1.0, 1.0
    for k, v in zip("xyz", front):
        cam["front"][k] = v
    for k, v in zip("xyz", up):
        cam["up"][k] = v
    cam["fov"] = f32(f32(f32(75.0) * f32(3.1415)) / f32(180.0))
    cam["aspect_ratio"] = f32(f32(width) / f32(height))
    cam["aperture"] = 0.0
    cam["focus_distance"] = 10.0
    return cam


def _strip(a):
    """Field-wise view of a structured array without padding members (the
    reference leaves float3 padding uninitialised, mathlib.hpp:74-76)."""
    import numpy.lib.recfunctions as rfn
    out = []

    def walk(arr, dt):
        for name in dt.names:
            sub = dt.fields[name][0]
            if name in ("w", "padding", "pad"):
                continue
            if sub.names:
                walk(arr[name], sub)
            else:
                out.append(np.ascontiguousarray(arr[name]).reshape(len(arr), -1).view(np.uint32))
    walk(a, a.dtype)
    return np.concatenate(out, axis=1) if out else np.zeros((len(a), 0), np.uint32)


def records_equal(a, b):
    """Bitwise equality of two record arrays over their payload fields."""
    if a.dtype != b.dtype or a.shape != b.shape:
        return False
    if a.size == 0:
        return True
    if a.dtype.names is None:
        return np.array_equal(a.view(np.uint8), b.view(np.uint8))
    return np.array_equal(_strip(a), _strip(b))
