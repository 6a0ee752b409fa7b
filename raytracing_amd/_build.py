"""Build recipes for the native parts (called by __graft_entry__.build()).

  librt_hip.so   HIP kernels + C-ABI, hipcc --offload-arch=gfx950 (cross-compiles without a GPU)
  librt_host.so  C++ host layer (Scene, Bvh, Integrator, HIPPathTraceIntegrator, Render) + flat C API
  rt_render      headless CLI with the reference's flags
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
INC = ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-I" + HOST]

# -ffp-contract=off + correctly rounded divide/sqrt + no fast-math: the
# arithmetic contract that makes GPU results bit-identical to the CPU oracle.
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
             "-fhip-fp32-correctly-rounded-divide-sqrt", "-fPIC", "-shared", "-ldl"]
CXX_FLAGS = ["-std=c++17", "-O2", "-ffp-contract=off", "-fPIC"]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    subprocess.check_call(cmd, cwd=ROOT)


def hipcc():
    for c in ("hipcc", "/opt/rocm/bin/hipcc"):
        try:
            subprocess.check_output([c, "--version"], stderr=subprocess.STDOUT)
            return c
        except Exception:
            continue
    raise RuntimeError("hipcc not found")


# librt_hip.so = three translation units (round 6; rt_hip.hip alone was 3 600 lines): the C-ABI with the hot path's kernels, the tree work on the device
# (a code object of its own: the hot path's is not rebuilt or re-hashed when the builder changes), and the host side of the tree work.
HIP_UNITS = [
    ("rt_hip", "rt_hip.hip", True),            # C-ABI, frames, launch logic + kernels.h (the hot path)
    ("device_fold", "device_fold.hip", True),  # fold_kernels.h + its host driver
    ("wide_bvh", "wide_bvh.cpp", False),       # build_wide_bvh, pair layout, the adaptation's host walks (no device code)
]


def _unit_sources(src):
    """what a unit is rebuilt for: its source and every header of csrc/ + include/ (coarse, like the single-unit build was)"""
    srcs = [os.path.join(CSRC, src)] + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    return srcs + [os.path.join(ROOT, "include", f) for f in ("rt_hip.h", "rt_types.h")]


def build_hip(force=False, extra_flags=(), out=None, obj_dir=None):
    out = out or os.path.join(HERE, "librt_hip.so")
    obj_dir = obj_dir or os.path.join(HERE, "build")
    os.makedirs(obj_dir, exist_ok=True)
    compile_flags = [f for f in HIP_FLAGS if f not in ("-shared", "-ldl")] + ["-c"] + list(extra_flags)
    objs, procs = [], []
    for name, src, device in HIP_UNITS:
        obj = os.path.join(obj_dir, name + ".o")
        objs.append(obj)
        if force or _newer(obj, _unit_sources(src)):
            flags = compile_flags if device else [f for f in compile_flags if not f.startswith("--offload-arch") and not f.startswith("-fhip")]
            procs.append(subprocess.Popen([hipcc()] + flags + INC + [os.path.join(CSRC, src), "-o", obj], cwd=ROOT))
    failed = [p.args for p in procs if p.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    if force or _newer(out, objs):
        _run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", out])
    return out


def build_host(force=False):
    out = os.path.join(HERE, "librt_host.so")
    cpps = [os.path.join(HOST, f) for f in ("bvh.cpp", "scene.cpp", "scene_cache.cpp", "png_loader.cpp", "jpeg_loader.cpp", "integrator.cpp",
                                            "hip_pt_integrator.cpp", "render.cpp", "host_capi.cpp")]
    deps = cpps + [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".hpp")]
    deps += [os.path.join(ROOT, "include", f) for f in ("rt_hip.h", "rt_types.h")]
    hip = os.path.join(HERE, "librt_hip.so")
    if force or _newer(out, deps + [hip]):
        _run(["g++"] + CXX_FLAGS + ["-shared"] + INC + cpps + ["-o", out, "-L" + HERE, "-lrt_hip", "-lz", "-pthread", "-Wl,-rpath,$ORIGIN"])
    exe = os.path.join(HERE, "rt_render")
    main = os.path.join(HOST, "main.cpp")
    if force or _newer(exe, [main, out]):
        _run(["g++"] + CXX_FLAGS[:-1] + INC + [main, "-o", exe, "-L" + HERE, "-lrt_host", "-lrt_hip",
                                               "-Wl,-rpath,$ORIGIN"])
    return out


def build_oracle():
    """The CPU checkers (test infrastructure): oracle/liboracle.so always,
    oracle/_ref/*.so when the reference checkout is present."""
    _run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    if os.path.isdir("/root/reference/src/kernels/cl"):
        _run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])


def build_all(force=False):
    build_hip(force)
    build_host(force)
    build_oracle()
