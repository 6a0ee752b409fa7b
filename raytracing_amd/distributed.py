"""Multi-GPU tiling: one process per GPU (torch.distributed, backend "nccl" =
RCCL over xGMI), the image split into interleaved row bands, scene replicated,
NO communication while rendering, and exactly ONE gather of the accumulated
radiance to rank 0 at the end.

Pixels are independent (every path is keyed by its global pixel coordinates),
so the assembled image is bit-identical to a single-GPU render."""
import numpy as np


def tile_rows(height, rank, world, band_height=8):
    """Global rows owned by `rank`: bands of `band_height` rows dealt round-robin
    (same rule as rt_frame_desc / rt_frame_global_row in include/rt_hip.h)."""
    rows = []
    band = rank
    while band * band_height < height:
        start = band * band_height
        rows.extend(range(start, min(start + band_height, height)))
        band += world
    return np.asarray(rows, dtype=np.int64)


def max_tile_rows(height, world, band_height=8):
    return max(len(tile_rows(height, r, world, band_height)) for r in range(world))


def gather_image(local, height, width, rank, world, band_height=8, dst=0):
    """One collective: gathers the per-rank tiles (torch tensors [rows, width, 4],
    on the device the process group uses) to `dst` and un-interleaves them.
    Returns the full [height, width, 4] tensor on `dst`, None elsewhere.
    Tiles are padded to the largest tile so the gather is a single fixed-size
    ncclGather (RCCL) -- 16 B per pixel, point-to-point over xGMI."""
    import torch
    import torch.distributed as dist
    pad_rows = max_tile_rows(height, world, band_height)
    send = torch.zeros((pad_rows, width, 4), dtype=torch.float32, device=local.device)
    send[: local.shape[0]] = local
    if world == 1:
        parts = [send]
    else:
        parts = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
        dist.gather(send, parts, dst=dst)
    if rank != dst:
        return None
    full = torch.empty((height, width, 4), dtype=torch.float32, device=local.device)
    for r in range(world):
        rows = torch.as_tensor(tile_rows(height, r, world, band_height), device=local.device)
        full[rows] = parts[r][: len(rows)]
    return full
