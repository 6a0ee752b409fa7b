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


def build_hip(force=False):
    out = os.path.join(HERE, "librt_hip.so")
    srcs = [os.path.join(CSRC, "rt_hip.hip")] + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    srcs += [os.path.join(ROOT, "include", f) for f in ("rt_hip.h", "rt_types.h")]
    if force or _newer(out, srcs):
        _run([hipcc()] + HIP_FLAGS + INC + [srcs[0], "-o", out])
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
