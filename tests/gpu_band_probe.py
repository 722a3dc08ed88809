import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rayca_amd import Config, DeviceScene, flatten, scenes, abi
W, H = 1920, 1080
cfg = Config(max_depth=1)
ds = DeviceScene(flatten(scenes.atrium_scene()), cfg, builder=abi.BUILDER_SAH)
dev = torch.device("cuda", 0)
out = torch.empty((H, W, 4), dtype=torch.uint8, device=dev)
for _ in range(12): ds.render_device(cfg, W, H, out.data_ptr(), 0, want_stats=True)
for parts, sel in ((135, (0, 20, 40, 60, 67, 80, 100, 120, 134)), (8, (0, 3, 7)), (1, (0,))):
    for part in sel:
        v = [ds.render_device(cfg, W, H, out.data_ptr(), 0, tile=(part, parts, 8), want_stats=True)["trace_kernel_ms"] for _ in range(8)][2:]
        print(f"parts {parts:3d} part {part:3d}: k_generation {np.median(v)*1000:7.1f} us", flush=True)
