"""Load balance of the row split over 8 ranks: kernel time of every rank's share for bands of 8 rows dealt round-robin (what
bench.py does) against one contiguous block of rows per rank (no de-interleave after the gather).  usage: python tests/gpu_band_balance_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from rayca_amd import Config, DeviceScene, flatten, scenes, abi
dev = torch.device("cuda", 0)
cfg, W, H = Config(max_depth=1), 1920, 1080
ds = DeviceScene(flatten(scenes.atrium_scene()), cfg, builder=abi.BUILDER_SAH)
ds.finish()
out = torch.empty((H, W, 4), dtype=torch.uint8, device=dev)
st = torch.cuda.Stream(dev)
for _ in range(16):
    ds.render_device(cfg, W, H, out.data_ptr(), 0, stream=st.cuda_stream, want_stats=True)
for parts in (8, 4, 2):
    for band in (8, 16, 32, (H + parts - 1) // parts):
        ms = []
        for part in range(parts):
            t = [ds.render_device(cfg, W, H, out.data_ptr(), 0, tile=(part, parts, band), stream=st.cuda_stream, want_stats=True)["kernel_ms"] for _ in range(5)]
            ms.append(float(np.median(t)))
        print(f"parts {parts} band {band:4d}: slowest {max(ms):.4f} ms, mean {np.mean(ms):.4f}, fastest {min(ms):.4f}  (slowest / mean {max(ms) / np.mean(ms):.2f})", flush=True)
