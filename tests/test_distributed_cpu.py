"""N>1 path on CPU: two gloo ranks shard the rows of a frame with the product's tiling helpers,
gather once, and rank 0 must hold the exact full frame.  The tile pixels come from the oracle here
(there is no GPU in this container); on the GPU box test_gpu_parity.py checks the same reassembly
with tiles rendered by the HIP kernels."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, height, width, band, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as ol
    from rayca_amd import Config, flatten, scenes
    from rayca_amd.distributed import gather_frame, rows_of, tile_of
    cfg = Config(max_depth=1)
    orc = ol.OracleScene(flatten(scenes.cornell_scene()), cfg, threads=2)
    tile = tile_of(rank, world, band)
    u8, _, st = orc.render(cfg, width, height, tile=tile, want_f32=False)
    assert st["rows_rendered"] == rows_of(tile, height).numel() == u8.shape[0]
    frame = gather_frame(torch.from_numpy(u8), height, band)
    # the batched form bench.py uses for N > 1: three frames (the frame, its complement, zeros) in one collective
    from rayca_amd.distributed import FrameGatherer
    g3 = FrameGatherer(height, width, band, "cpu", batch=3)
    send = g3.new_send()
    send[0, : g3.my_rows] = torch.from_numpy(u8)
    send[1, : g3.my_rows] = 255 - torch.from_numpy(u8)
    batch = g3.gather_batch(send)
    g1 = FrameGatherer(height, width, band, "cpu", batch=1)   # --gather-batch 1: the same call shape, one frame
    send1 = g1.new_send()
    send1[0, : g1.my_rows] = torch.from_numpy(u8)
    one = g1.gather_batch(send1)
    assert (one is None) == (rank != 0)
    # the call shape of bench.py's per-frame gather (StreamGatherLoop): the frame is rendered straight into a full-size
    # (max_rows) send buffer, ragged heights included, and gathered with out= a buffer that is reused
    gs = FrameGatherer(height, width, band, "cpu")
    full_send = torch.zeros((gs.max_rows, width, 4), dtype=torch.uint8)
    full_send[: gs.my_rows] = torch.from_numpy(u8)
    reuse = torch.empty((height, width, 4), dtype=torch.uint8) if rank == 0 else None
    for _ in range(2):
        streamed = gs(full_send, out=reuse)
    assert (streamed is None) == (rank != 0)
    if rank == 0:
        assert streamed is reuse and np.array_equal(streamed.numpy(), frame.numpy())
        full, _, _ = orc.render(cfg, width, height, want_f32=False)
        assert np.array_equal(one[0].numpy(), frame.numpy())
        assert np.array_equal(batch[0].numpy(), frame.numpy())
        assert np.array_equal(batch[1].numpy(), 255 - frame.numpy())
        assert int(batch[2].sum()) == 0
        np.save(out_path, np.stack([frame.numpy(), full]))
    else:
        assert frame is None and batch is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("height,band", [(64, 8), (45, 8), (37, 4)])
def test_two_rank_gather_reassembles_frame(tmp_path, height, band):
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, _free_port(), height, 80, band, out), nprocs=2, join=True)
    got, want = np.load(out)
    assert np.array_equal(got, want)
    assert got[..., :3].sum() > 0


def test_rows_of_matches_library(product_lib):
    import ctypes as C
    from rayca_amd import abi
    from rayca_amd.distributed import rows_of, tile_of
    t = abi.RaycaTile()
    for h in (1, 9, 1080):
        for world in (1, 2, 4, 8):
            allrows = []
            for r in range(world):
                rows = rows_of(tile_of(r, world, 8), h)
                t.part, t.parts, t.band_rows = r, world, 8
                assert product_lib.rayca_hip_tile_rows(C.byref(t), h) == rows.numel()
                allrows += rows.tolist()
            assert sorted(allrows) == list(range(h))
