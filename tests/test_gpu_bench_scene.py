"""-m gpu: bench.py on an asset FROM DISK (row g of VERDICT r02: the headline on the named scenes is one flag away --
`bench.py --scene exterior.obj --flip-yz --scale 0.01`, run_bistro.bat:15 -- and that flag has to work the day the asset is
there).  The asset here is generated: an OBJ + MTL whose materials reference PNG (one of them Adam7-interlaced), JPEG and
TGA textures, loaded by the C++ loader (host/scene.cpp, src/scene/scene.cpp:127-322), rendered by the same timed region as
the stand-ins; the line must say data: "real" and carry every object of the contract."""
import json
import os
import shutil
import subprocess
import sys
import numpy as np
import pytest
from raytracing_amd import scenes as S
from tests.test_host_layer import _write_png

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_py_renders_a_scene_file_with_textures(tmp_path):
    obj = S.shader_balls_obj(str(tmp_path), 3000)
    rng = np.random.RandomState(11)
    _write_png(str(tmp_path / "albedo.png"), rng.randint(40, 256, (32, 48, 3)), 2, filters=[4, 1, 2], interlace=True)
    _write_png(str(tmp_path / "rough.png"), rng.randint(0, 256, (16, 16, 1)), 0, filters=[0, 3])
    shutil.copy(os.path.join(ROOT, "tests", "golden", "jpeg", "baseline_420_restart.jpg"), tmp_path / "photo.jpg")
    hdr = bytes([0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 8, 0, 8, 0, 24, 0x20])
    (tmp_path / "spec.tga").write_bytes(hdr + bytes(rng.randint(0, 256, 8 * 8 * 3).astype(np.uint8)))
    mtl = (tmp_path / "ShaderBalls.mtl").read_text()
    mtl = mtl.replace("newmtl floor", "newmtl floor\nmap_Kd albedo.png", 1).replace("newmtl mat00", "newmtl mat00\nmap_Kd photo.jpg\nmap_Ks spec.tga\nmap_Pr rough.png", 1)
    assert "albedo.png" in mtl and "photo.jpg" in mtl
    (tmp_path / "ShaderBalls.mtl").write_text(mtl)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--scene", obj, "--width", "480", "--height", "270", "--bounces", "4",
                        "--steps", "1", "--warmup", "1", "--samples-per-step", "8", "--per-frame-frames", "4", "--no-cpu-baseline"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["data"] == "real" and line["n_gpus"] == 1 and line["value"] > 0 and line["unit"] == "Mrays/s"
    assert "scene file ShaderBallsStandIn.obj" in line["config"]["workload"]
    assert line["config"]["triangles"] > 20000 and line["config"]["scene_s"] >= 0 and line["config"]["setup_s"] >= line["config"]["scene_s"]
    assert line["per_frame"]["mrays_per_s"] > 0 and line["per_frame"]["frames"] == 4
    assert line["roofline"]["bound"] == "hbm" and line["roofline"]["live"]["rays_per_launch"] > 0
    assert line["config"]["non_finite_pixels"] <= 0.01 * 480 * 270      # the mirror balls' inf * 0 (material.h:79-81,230), as in the reference


def test_bench_py_two_tiles_gather_the_reference_kernels_frame(tmp_path):
    """bench.py --gpus 2 the way the driver launches it, on the ONE GPU a test box has (--debug-shared-gpu: both ranks on
    device 0, the gather over gloo instead of RCCL): two processes, interleaved bands, scene built once and loaded from
    rank 0's cache, one line -- whose `parity` says that the frame gathered from the two tiles is, bit for bit, the frame
    the reference's own kernels render (the leg bench.py runs at every N > 1)."""
    from tests import _ref
    if not _ref.available():
        pytest.skip("oracle/_ref is not built here")
    obj = S.shader_balls_obj(str(tmp_path), 2000)
    for extra in ([], ["--debug-try-rccl"]):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--debug-shared-gpu", "--scene", obj, "--width", "320",
                            "--height", "200", "--bounces", "5", "--steps", "1", "--warmup", "1", "--samples-per-step", "8"] + extra,
                           cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "strong"
        assert len(line["ranks"]["render_ms"]) == 2 and sum(line["ranks"]["rows"]) == 200
        p = line["parity"]
        assert p["tiles"] == 2 and p["bit_identical"] is True and p["differing_pixels"] == 0, p
        assert line["cpu_baseline"] is None                             # timed at N = 1 only
        # --debug-try-rccl: RCCL refuses two ranks on one device; every rank then agrees on the gloo gather and the line says why
        assert ("RCCL FALLBACK" in line["gather"]["transport"]) == bool(extra), line["gather"]
        # one fold adaptation per group (round 6): rank 1 uploaded without a shadow tree / an adaptation and took rank 0's records
        assert line["ranks"]["fold_share"]["records"][0] > 0, line["ranks"]


def test_bench_py_eight_ranks_share_one_gpu_and_one_adaptation(tmp_path):
    """The driver's N = 8 launch on the one GPU of a test box (--debug-shared-gpu): eight processes, eight tiles of interleaved bands, rank 0's scene cache,
    ONE fold adaptation for the group -- rank 0's records broadcast to the seven others, which uploaded without a shadow tree or an adaptation of their own
    (their set-up is the cheap part: `setup_breakdown`) -- and the frame gathered from the eight tiles is the reference kernels' frame bit for bit."""
    from tests import _ref
    if not _ref.available():
        pytest.skip("oracle/_ref is not built here")
    obj = S.shader_balls_obj(str(tmp_path), 2000)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--debug-shared-gpu", "--scene", obj, "--width", "320",
                        "--height", "200", "--bounces", "4", "--steps", "1", "--warmup", "1", "--samples-per-step", "4"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 8 and line["value"] > 0
    assert len(line["ranks"]["render_ms"]) == 8 and sum(line["ranks"]["rows"]) == 200
    assert line["ranks"]["fold_share"]["records"][0] > 0
    p = line["parity"]
    assert p["tiles"] == 8 and p["bit_identical"] is True and p["differing_pixels"] == 0, p
