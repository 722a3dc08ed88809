"""Frames in flight: one scene, one frame context + stream per frame in flight, frames dealt round-robin.  Does the tail of frame i overlap the
head of frame i+1?  usage: python tests/gpu_inflight_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rayca_amd import Config, DeviceScene, IntegratorStrategy, flatten, scenes, abi
W, H = 1920, 1080
cfg = Config(max_depth=1) if (len(sys.argv) < 2 or sys.argv[1] != 'flat') else Config(integrator=IntegratorStrategy.Flat)
desc = flatten(scenes.atrium_scene())
dev = torch.device("cuda", 0)
NH = 8
from rayca_amd.streams import frame_streams
streams = frame_streams(dev, NH, spare=0)[0]   # consecutive streams on consecutive hardware queues (rayca_amd/streams.py)
ds = DeviceScene(desc, cfg, builder=abi.BUILDER_SAH)   # one scene, NH frame contexts
ds.finish()
for parts in (1, 2, 4, 8):
    tile = (0, parts, 8)
    rows = ds.tile_rows(tile, H)
    outs = [torch.empty((rows, W, 4), dtype=torch.uint8, device=dev) for _ in range(NH)]
    for i, (s, o) in enumerate(zip(streams, outs)):
        for _ in range(8):
            ds.render_device(cfg, W, H, o.data_ptr(), 0, tile=tile, stream=s.cuda_stream, want_stats=True, context=i)
    torch.cuda.synchronize()
    for inflight in (1, 2, 3, 4, 6, 8):
        K = 300
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            j = i % inflight
            ds.render_device(cfg, W, H, outs[j].data_ptr(), 0, tile=tile, stream=streams[j].cuda_stream, context=j)
        t_host = (time.perf_counter() - t0) / K * 1e3   # the host's share: enqueueing one frame
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / K * 1e3
        print(f"parts {parts} frames in flight {inflight}: {t:.4f} ms/frame (host enqueue {t_host:.4f})", flush=True)
