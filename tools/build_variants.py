"""Compile-time A/B variants of librt_hip.so for experiments on the GPU box: each variant is the same source built with
extra -D macros into raytracing_amd/variants/<name>/librt_hip.so (git-ignored like every built .so, shipped by gpurun).
A call script swaps one in with `cp raytracing_amd/variants/<name>/librt_hip.so raytracing_amd/librt_hip.so` (the box
works on a scratch copy of the tree).  usage: python tools/build_variants.py name=-DMACRO[,-DMACRO2] ...
The production source carries no experiment macros (round 3): the hooks round 2's sensitivity runs used (RT_W4_EXTRA_ACCESS,
RT_W4_EXTRA_VALU, RT_SHADE_EXTRA_*, RT_SHADE_LDS_PAD) are kept as profiles/experiments/r02_sensitivity_hooks.diff -- apply it
to a scratch copy first; RT_SHADE_BLOCK (k_shade's block size) is the one macro the source still honours."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import _build
for spec in sys.argv[1:]:
    name, macros = spec.split("=", 1)
    d = os.path.join(ROOT, "raytracing_amd", "variants", name)
    os.makedirs(d, exist_ok=True)
    try:
        _build.build_hip(force=True, extra_flags=macros.split(","), out=os.path.join(d, "librt_hip.so"), obj_dir=os.path.join(d, "build"))
        print(name, "ok")
    except Exception as e:      # noqa: BLE001
        print(name, "FAILED", e)
