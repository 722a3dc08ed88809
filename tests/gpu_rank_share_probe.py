"""What one rank of an N-GPU run costs per frame on THIS GPU: render part 0 of `parts` (bands of 8 rows) with F frames in
flight, optionally followed by a per-frame RCCL gather on a 1-rank group (the collective's launch + Python cost without the
wire and without the other ranks' skew).  usage: python tests/gpu_rank_share_probe.py [parts ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np
import torch
import torch.distributed as dist
from rayca_amd import Config, DeviceScene, flatten, scenes, abi
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
cfg, W, H = Config(max_depth=1), 1920, 1080
# RAYCA_PROBE_STREAMS=first|last|naive: the frame streams from rayca_amd.streams.frame_streams before the scene is made / after
# it, or the first torch streams asked for after the scene (how the probe used to do it)
from rayca_amd.streams import frame_streams
MODE = os.environ.get("RAYCA_PROBE_STREAMS", "first")
early_streams = None
if MODE == "first":
    fs, sp = frame_streams(dev, 8)
    early_streams = fs + sp
_lib = None
if os.environ.get("RAYCA_PROBE_LIB"):   # a library variant from rayca_amd/csrc/variants (tests/build_variants.sh)
    import ctypes
    _lib = abi.bind_product_signatures(ctypes.CDLL(os.path.join(ROOT, "rayca_amd", "csrc", "variants", f"librayca_{os.environ['RAYCA_PROBE_LIB']}.so")))
ds = DeviceScene(flatten(scenes.atrium_scene()), cfg, builder=abi.BUILDER_SAH, **({"_lib": _lib} if _lib else {}))
ds.finish()
if MODE == "last":
    fs, sp = frame_streams(dev, 8)
    early_streams = fs + sp
parts_list = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
whole_ms = None
for parts in parts_list:
    tile = (0, parts, 8)
    rows = ds.tile_rows(tile, H)
    for F in [int(x) for x in os.environ.get("RAYCA_PROBE_F", "2,4").split(",")]:
        streams = early_streams[8 - F:8] if early_streams else [torch.cuda.Stream(dev) for _ in range(F)]
        comm = early_streams[8] if early_streams else torch.cuda.Stream(dev)
        sends = [torch.empty((rows, W, 4), dtype=torch.uint8, device=dev) for _ in range(F)]
        recv = [torch.empty((rows, W, 4), dtype=torch.uint8, device=dev)]
        ev = [torch.cuda.Event() for _ in range(F)]
        for i in range(F):   # contexts warm, node format decided
            for _ in range(10):
                ds.render_device(cfg, W, H, sends[i].data_ptr(), 0, tile=tile, stream=streams[i].cuda_stream, context=i, want_stats=True)
        issue = [ds.prepare_device(cfg, W, H, sends[i].data_ptr(), 0, tile=tile, stream=streams[i].cuda_stream, context=i) for i in range(F)]
        for gather in (False, True):
            K = 200
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(K):
                i = k % F
                issue[i]()
                ev[i].record(streams[i])
                if gather:
                    with torch.cuda.stream(comm):
                        comm.wait_event(ev[i])
                        dist.gather(sends[i], recv, dst=0)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / K * 1e3
            if parts == 1 and not gather:
                whole_ms = ms   # one GPU's own frame time, measured here: what the shares are compared with
            scale = f"x{whole_ms / ms:.2f} of one GPU's" if whole_ms else "(run with parts 1 first for the ratio)"
            print(f"parts {parts} ({rows} rows) F={F} gather={'per frame' if gather else 'none':9s}: {ms:.4f} ms/frame  -> {parts}-GPU frame rate {scale}", flush=True)
dist.destroy_process_group()
