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
