"""What tile-parallel scaling can be at best: one rank's share of the frame (part 0 of N) rendered back to back,
without the gather.  ms/step against N shows the fixed per-frame cost (launches, work-queue reset, resolve) that
strong scaling of a 0.67 ms frame runs into.  usage: python tests/gpu_scaling_probe.py [workload]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rayca_amd import Config, DeviceScene, flatten, scenes, abi
wl = sys.argv[1] if len(sys.argv) > 1 else "atrium"
W, H = (1920, 1080)
cfg = Config(max_depth=1)
ds = DeviceScene(flatten(scenes.atrium_scene()), cfg, builder=abi.BUILDER_SAH)
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(dev)
base = None
for parts in (1, 2, 4, 8):
    tile = (0, parts, 8)
    rows = ds.tile_rows(tile, H)
    out = torch.empty((rows, W, 4), dtype=torch.uint8, device=dev)
    for _ in range(20):
        ds.render_device(cfg, W, H, out.data_ptr(), 0, tile=tile, stream=stream.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 300
    for _ in range(K):
        ds.render_device(cfg, W, H, out.data_ptr(), 0, tile=tile, stream=stream.cuda_stream)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / K * 1e3
    st = ds.render_device(cfg, W, H, out.data_ptr(), 0, tile=tile, stream=stream.cuda_stream, want_stats=True)
    base = base or t
    print(f"parts {parts}: {t:.4f} ms/step (host enqueue {t_host / K * 1e3:.4f} ms), kernels {st['kernel_ms']:.4f} ms, speedup {base / t:.2f}x", flush=True)
