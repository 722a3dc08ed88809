"""Multi-GPU frame assembly: image rows shard across ranks, one gather at frame end.

Pixels are independent and the scene is read-only, so every rank holds the whole scene + BVH and
renders its own rows (RaycaTile: bands of `band_rows` rows dealt round-robin, which balances sky
against geometry).  The only exchange is the frame-end gather of RGBA8 rows to rank 0 over
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests):
4 B/pixel, 33 MB for 3840x2160 -- latency-bound, not bandwidth-bound.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def tile_of(rank: int, world: int, band_rows: int = 8):
    """(part, parts, band_rows) as RaycaTile wants it."""
    return (rank, world, band_rows)


def rows_of(tile, height: int) -> torch.Tensor:
    """Frame rows rendered by `tile`, ascending (matches rayca_hip_tile_rows / the kernels' row map)."""
    part, parts, band = tile
    y = torch.arange(height)
    if parts <= 1:
        return y
    return y[((y // max(band, 1)) % parts) == part]


def gather_frame(local_rows: torch.Tensor, height: int, band_rows: int, dst: int = 0, group=None):
    """local_rows: (rows_of_this_rank, W, 4) uint8 on this rank's device.  Returns the assembled
    (height, W, 4) frame on rank `dst`, None elsewhere.  One collective per frame."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1:
        return local_rows
    width = local_rows.shape[1]
    counts = [int(rows_of(tile_of(r, world, band_rows), height).numel()) for r in range(world)]
    max_rows = max(counts)
    send = local_rows
    if send.shape[0] != max_rows:  # ragged heights: pad so every rank contributes the same count
        pad = torch.zeros((max_rows - send.shape[0], width, 4), dtype=send.dtype, device=send.device)
        send = torch.cat([send, pad], 0)
    send = send.contiguous()
    recv = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, recv, dst=dst, group=group)
    if rank != dst:
        return None
    frame = torch.empty((height, width, 4), dtype=send.dtype, device=send.device)
    for r in range(world):
        idx = rows_of(tile_of(r, world, band_rows), height).to(send.device)
        frame.index_copy_(0, idx, recv[r][: counts[r]])
    return frame
