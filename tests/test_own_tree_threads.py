"""own_bvh.h builds the shadow rays' own tree on a pool of host threads (round 6: the top of the tree too).  tests/native/own_bvh_threads.cpp checks, on clustered,
flat and coincident-centroid leaf sets of 2 .. 150 000 leaves, that 2, 3, 8 and 32 threads build byte for byte the tree one thread builds, and that a cancel flag
raised before or during a build ends it (false, no tree) -- here under ThreadSanitizer and under AddressSanitizer + UBSan.  The header is plain C++: g++ alone."""
import os
import shutil
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
@pytest.mark.parametrize("sanitizer,leaves", [("thread", "70000"), ("address,undefined", "150000")])
def test_the_tree_does_not_depend_on_the_thread_count(tmp_path, sanitizer, leaves):
    exe = str(tmp_path / "own_bvh_threads")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-pthread", "-fsanitize=" + sanitizer, "-fno-sanitize-recover=all", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "raytracing_amd", "csrc"), os.path.join(ROOT, "tests", "native", "own_bvh_threads.cpp"), "-o", exe]
    build = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("this g++ has no %s sanitizer runtime: %s" % (sanitizer, build.stderr[-200:]))
    assert build.returncode == 0, build.stderr[-2000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1")
    run = subprocess.run([exe, leaves], capture_output=True, text=True, timeout=900, env=env)
    if run.returncode != 0 and "FATAL: ThreadSanitizer: unexpected memory mapping" in run.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow in this container")
    assert run.returncode == 0 and "ok: 14 trees identical" in run.stdout, (run.stdout[-500:], run.stderr[-3000:])


@pytest.mark.skipif(shutil.which("g++") is None or not os.path.exists("/opt/rocm/include/hip/hip_vector_types.h"), reason="no g++ or no HIP headers")
@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_the_host_fold_does_not_depend_on_the_thread_count(tmp_path, sanitizer):
    """build_wide_bvh (wide_bvh.cpp; round 6: validation in slices, the dynamic programme over index ranges, record numbering by subtrees): with measured weights,
    the surface area, a metric and the two-level collapse, 2 / 5 / 16 threads give the one-thread records, order and roots; an array that is a tree but not in
    depth-first layout makes the range sweep fall back (tests/native/wide_bvh_threads.cpp)."""
    exe = str(tmp_path / "wide_bvh_threads")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-pthread", "-D__HIP_PLATFORM_AMD__", "-fsanitize=" + sanitizer, "-fno-sanitize-recover=all", "-I", "/opt/rocm/include",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "raytracing_amd", "csrc"), os.path.join(ROOT, "tests", "native", "wide_bvh_threads.cpp"),
           os.path.join(ROOT, "raytracing_amd", "csrc", "wide_bvh.cpp"), "-o", exe]
    build = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("this g++ has no %s sanitizer runtime: %s" % (sanitizer, build.stderr[-200:]))
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe, "140000"], capture_output=True, text=True, timeout=900, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    if run.returncode != 0 and "ThreadSanitizer: unexpected memory mapping" in run.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow in this container")
    assert run.returncode == 0 and "ok: 13 folds identical" in run.stdout, (run.stdout[-500:], run.stderr[-3000:])


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_the_record_order_does_not_depend_on_the_thread_count(tmp_path, sanitizer):
    """treelet_order.h (rt_scene_upload's record order of the exact BVH2; round 6: the clusters below depth 4 ordered by a pool): treelet sizes 1 / 3 / 7 / 15 on
    2 / 5 / 16 threads give the one-thread order; a shared interior child and a child index behind the array are refused on every thread count
    (tests/native/treelet_order_threads.cpp)."""
    exe = str(tmp_path / "treelet_order_threads")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-pthread", "-fsanitize=" + sanitizer, "-fno-sanitize-recover=all", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "raytracing_amd", "csrc"), os.path.join(ROOT, "tests", "native", "treelet_order_threads.cpp"), "-o", exe]
    build = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("this g++ has no %s sanitizer runtime: %s" % (sanitizer, build.stderr[-200:]))
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe, "150000"], capture_output=True, text=True, timeout=900, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    if run.returncode != 0 and "ThreadSanitizer: unexpected memory mapping" in run.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow in this container")
    assert run.returncode == 0 and "ok: 14 orders identical" in run.stdout, (run.stdout[-500:], run.stderr[-3000:])
