"""The N>1 path on CPU: world_size-2 `gloo` process group; every rank owns the
interleaved row bands bench.py would render on its GPU, fills them from the CPU
oracle's image (tiling invariance of the HIP path itself is a -m gpu test), and
ONE gather assembles the frame on rank 0."""
import os
import socket
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from raytracing_amd import distributed as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, height, width, band, image_path, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = torch.from_numpy(np.load(image_path))
    rows = D.tile_rows(height, rank, world, band)
    local = full[torch.as_tensor(rows)].clone()              # what this rank's GPU tile would hold
    got = D.gather_image(local, height, width, rank, world, band)
    if rank == 0:
        np.save(out_path, got.numpy())
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("height,band", [(64, 8), (50, 8), (33, 4)])
def test_two_rank_gather_reassembles_the_frame(tmp_path, height, band, golden_scenes):
    from tests import _oracle
    from raytracing_amd import types as T
    width = 48
    orc = _oracle.Oracle(width, height, golden_scenes["cornell"])
    orc.set_camera(T.default_camera(width, height))
    orc.set_max_bounces(2)
    orc.integrate(1)
    img = orc.radiance()
    np.save(tmp_path / "img.npy", img)
    port = _free_port()
    mp.spawn(_worker, args=(2, port, height, width, band, str(tmp_path / "img.npy"), str(tmp_path / "out.npy")),
             nprocs=2, join=True)
    assert np.array_equal(np.load(tmp_path / "out.npy"), img)


def test_tile_rows_partition_every_row_exactly_once():
    for height in (1, 7, 8, 9, 64, 720, 1080):
        for world in (1, 2, 3, 4, 8):
            for band in (1, 4, 8, 16):
                rows = np.concatenate([D.tile_rows(height, r, world, band) for r in range(world)])
                assert sorted(rows.tolist()) == list(range(height))
    # balance at the benchmark size: 720 rows, 8 ranks, bands of 8 -> 88..96 rows each
    sizes = [len(D.tile_rows(720, r, 8, 8)) for r in range(8)]
    assert max(sizes) - min(sizes) <= 8


def _bench(*args, timeout=300):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(args), capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout            # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_bench_launches_itself_for_n_gpus_and_reports_one_line():
    """`python bench.py --gpus 2` as the driver runs `--gpus 1`: no launcher, the script re-executes itself under
    torch.distributed.run, both ranks rendezvous, the tiles are gathered on rank 0 and ONE JSON line comes out.
    (--plumbing-only: everything except the rendering, which needs a GPU.)"""
    line = _bench("--gpus", "2", "--plumbing-only", "--width", "64", "--height", "50", "--steps", "3", "--warmup", "0")
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 0
    assert line["gather"]["nranks"] == 2 and line["gather"]["image_ok"] is True
    assert line["plumbing_only"] is True and line["value"] is None


def test_bench_under_an_external_launcher():
    """The driver's own launch line: torch.distributed.run sets RANK / WORLD_SIZE and bench.py must not launch again."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--plumbing-only", "--width", "48",
           "--height", "33", "--band-height", "4"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["gather"]["image_ok"] is True


def test_rt_render_plans_the_same_tiles_as_the_python_side():
    """`rt_render --gpus N --plan 1` prints TiledRender::TileRows -- the rows every GPU of the C++ multi-GPU path renders
    (rt_frame_desc's band rule) -- without touching a GPU: it must be the partition raytracing_amd.distributed.tile_rows
    (bench.py's gloo / RCCL assembly) uses, cover every row once, for even and ragged heights."""
    import subprocess
    exe = os.path.join(ROOT, "raytracing_amd", "rt_render")
    for gpus, height in ((2, 1080), (8, 1080), (3, 50), (8, 7), (5, 2160)):
        r = subprocess.run([exe, "--gpus", str(gpus), "-w", "64", "-h", str(height), "--plan", "1"], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stdout + r.stderr
        seen = []
        for rank in range(gpus):
            line = [l for l in r.stdout.splitlines() if l.startswith("tile %d of %d:" % (rank, gpus))][0]
            rows = [int(x) for x in line.split("=")[1].split()]
            assert rows == D.tile_rows(height, rank, gpus, 8).tolist()
            seen += rows
        assert sorted(seen) == list(range(height))
