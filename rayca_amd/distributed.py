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
    order.  Per frame that leaves one collective and one index_select launch on the caller's stream.

    batch > 1 gathers that many finished frames with ONE collective (fewer, larger messages: a rank's rows of a
    1080p frame are 1 MB at 8 ranks, far below what an xGMI link needs to reach its rate, and every collective is a
    synchronisation point of all ranks): the caller renders frame b of a batch straight into `send[b, :my_rows]` of
    a buffer from new_send() and calls gather_batch() once the batch is complete."""

    def __init__(self, height: int, width: int, band_rows: int, device, dst: int = 0, group=None, batch: int = 1):
        self.group, self.dst = group, dst
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.height, self.width, self.batch, self.device = height, width, max(1, int(batch)), device
        rows = [rows_of(tile_of(r, self.world, band_rows), height) for r in range(self.world)]
        self.counts = [int(r.numel()) for r in rows]
        self.max_rows = max(self.counts)
        self.my_rows = self.counts[self.rank]
        self.pad = None
        if self.my_rows != self.max_rows:  # ragged heights: every rank contributes the same count
            self.pad = torch.zeros((self.max_rows, width, 4), dtype=torch.uint8, device=device)
        self.recv = self.views = self.perm = None
        self.recv_b = self.views_b = self.perm_b = None
        if self.rank == dst and self.world > 1:
            self.recv = torch.empty((self.world, self.max_rows, width, 4), dtype=torch.uint8, device=device)
            self.views = [self.recv[r] for r in range(self.world)]
            # frame row y lives at gathered row perm[y] = rank * max_rows + position within that rank's rows
            perm = torch.empty(height, dtype=torch.int64)
            for r, ys in enumerate(rows):
                perm[ys] = r * self.max_rows + torch.arange(ys.numel())
            self.perm = perm.to(device)
            if self.batch > 1:
                self.recv_b = torch.empty((self.world, self.batch, self.max_rows, width, 4), dtype=torch.uint8, device=device)
                self.views_b = [self.recv_b[r] for r in range(self.world)]
                # row y of frame b lives at gathered row (rank * batch + b) * max_rows + position
                pb = torch.empty((self.batch, height), dtype=torch.int64)
                for b in range(self.batch):
                    for r, ys in enumerate(rows):
                        pb[b, ys] = (r * self.batch + b) * self.max_rows + torch.arange(ys.numel())
                self.perm_b = pb.reshape(-1).to(device)

    def __call__(self, local_rows: torch.Tensor, out: torch.Tensor = None):
        """local_rows: (rows_of_this_rank, W, 4) uint8.  Returns the (height, W, 4) frame on `dst`, None elsewhere."""
        if self.world == 1:
            return local_rows
        send = local_rows
        if self.pad is not None and local_rows.shape[0] != self.max_rows:   # (a caller may render straight into a full-size buffer)
            self.pad[: self.my_rows].copy_(local_rows)
            send = self.pad
        dist.gather(send, self.views, dst=self.dst, group=self.group)
        if self.rank != self.dst:
            return None
        if out is None:
            out = torch.empty((self.height, self.width, 4), dtype=torch.uint8, device=self.recv.device)
        torch.index_select(self.recv.view(self.world * self.max_rows, self.width, 4), 0, self.perm, out=out)
        return out

    def new_send(self) -> torch.Tensor:
        """(batch, max_rows, W, 4) buffer; frame b of a batch is rendered into [b, :my_rows] (the rows behind a
        rank's own count are padding that the de-interleave never reads)."""
        return torch.zeros((self.batch, self.max_rows, self.width, 4), dtype=torch.uint8, device=self.device)

    def gather_batch(self, send: torch.Tensor, out: torch.Tensor = None):
        """One collective for `batch` frames.  Returns (batch, height, W, 4) on `dst`, None elsewhere."""
        assert send.shape == (self.batch, self.max_rows, self.width, 4)
        if self.world == 1:
            return send[:, : self.my_rows]
        into = None
        if self.rank == self.dst:
            into = self.views_b if self.batch > 1 else [v.unsqueeze(0) for v in self.views]
        dist.gather(send, into, dst=self.dst, group=self.group)
        if self.rank != self.dst:
            return None
        if out is None:
            out = torch.empty((self.batch, self.height, self.width, 4), dtype=torch.uint8, device=self.device)
        if self.batch > 1:
            torch.index_select(self.recv_b.view(self.world * self.batch * self.max_rows, self.width, 4), 0, self.perm_b,
                               out=out.view(self.batch * self.height, self.width, 4))
        else:
            torch.index_select(self.recv.view(self.world * self.max_rows, self.width, 4), 0, self.perm, out=out.view(self.height, self.width, 4))
        return out


def gather_frame(local_rows: torch.Tensor, height: int, band_rows: int, dst: int = 0, group=None):
    """One-off form of FrameGatherer (tests, single frames)."""
    return FrameGatherer(height, local_rows.shape[1], band_rows, local_rows.device, dst, group)(local_rows.contiguous())
