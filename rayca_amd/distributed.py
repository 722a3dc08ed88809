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


class FrameGatherer:
    """Everything about the frame-end gather that does not change from frame to frame, set up once: per-rank row
    counts, the receive buffer on `dst`, and ONE permutation that de-interleaves the gathered bands into frame
    order.  Per frame that leaves one collective and one index_select launch on the caller's stream."""

    def __init__(self, height: int, width: int, band_rows: int, device, dst: int = 0, group=None):
        self.group, self.dst = group, dst
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.height, self.width = height, width
        rows = [rows_of(tile_of(r, self.world, band_rows), height) for r in range(self.world)]
        self.counts = [int(r.numel()) for r in rows]
        self.max_rows = max(self.counts)
        self.my_rows = self.counts[self.rank]
        self.pad = None
        if self.my_rows != self.max_rows:  # ragged heights: every rank contributes the same count
            self.pad = torch.zeros((self.max_rows, width, 4), dtype=torch.uint8, device=device)
        self.recv = self.views = self.perm = None
        if self.rank == dst and self.world > 1:
            self.recv = torch.empty((self.world, self.max_rows, width, 4), dtype=torch.uint8, device=device)
            self.views = [self.recv[r] for r in range(self.world)]
            # frame row y lives at gathered row perm[y] = rank * max_rows + position within that rank's rows
            perm = torch.empty(height, dtype=torch.int64)
            for r, ys in enumerate(rows):
                perm[ys] = r * self.max_rows + torch.arange(ys.numel())
            self.perm = perm.to(device)

    def __call__(self, local_rows: torch.Tensor, out: torch.Tensor = None):
        """local_rows: (rows_of_this_rank, W, 4) uint8.  Returns the (height, W, 4) frame on `dst`, None elsewhere."""
        if self.world == 1:
            return local_rows
        send = local_rows
        if self.pad is not None:
            self.pad[: self.my_rows].copy_(local_rows)
            send = self.pad
        dist.gather(send, self.views, dst=self.dst, group=self.group)
        if self.rank != self.dst:
            return None
        if out is None:
            out = torch.empty((self.height, self.width, 4), dtype=torch.uint8, device=self.recv.device)
        torch.index_select(self.recv.view(self.world * self.max_rows, self.width, 4), 0, self.perm, out=out)
        return out


def gather_frame(local_rows: torch.Tensor, height: int, band_rows: int, dst: int = 0, group=None):
    """One-off form of FrameGatherer (tests, single frames)."""
    return FrameGatherer(height, local_rows.shape[1], band_rows, local_rows.device, dst, group)(local_rows.contiguous())
