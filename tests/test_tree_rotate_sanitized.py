"""tree_rotate.h (RT_CTX_OPT_ADAPTIVE_FOLD bit 3) under AddressSanitizer + UBSan: tests/native/tree_rotate_fuzz.cpp builds 400 random, deliberately
lopsided binary trees over random leaf boxes (flat ones too), throws random rays at them (zero direction components, t_max 0, NaN origins), rotates with
every move set and checks after each run: the same leaves, exact-union boxes, the linear layout, a cost that did not rise -- and (round 6: the passes run on a
pool, disjoint subtrees on different threads) that 2 .. 8 threads with a hand-over grain of 1 .. 40 rays make the rotations one thread makes, byte for byte;
once more under ThreadSanitizer.  The header is plain C++, so g++ compiles it on its own."""
import os
import shutil
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
@pytest.mark.parametrize("sanitizer", ["address,undefined", "thread"])
def test_tree_rotations_under_the_sanitizers(tmp_path, sanitizer):
    exe = str(tmp_path / "tree_rotate_fuzz")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-pthread", "-fsanitize=" + sanitizer, "-fno-sanitize-recover=all", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "raytracing_amd", "csrc"), os.path.join(ROOT, "tests", "native", "tree_rotate_fuzz.cpp"), "-o", exe]
    build = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("this g++ has no sanitizer runtime: " + build.stderr[-200:])
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    if run.returncode != 0 and "ThreadSanitizer: unexpected memory mapping" in run.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow in this container")
    assert run.returncode == 0 and "ok: 400 random trees" in run.stdout, (run.stdout[-500:], run.stderr[-2000:])
