"""Does the frame's output format matter?  Device-resident frames, RGBA8 / RGBA32F / both, back to back.
usage: python tests/gpu_out_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rayca_amd import Config, DeviceScene, IntegratorStrategy, flatten, scenes, abi
desc = flatten(scenes.atrium_scene()); W, H = 1920, 1080
ds = DeviceScene(desc, Config(), builder=abi.BUILDER_SAH)
o8 = torch.empty(H * W * 4, dtype=torch.uint8, device="cuda"); o32 = torch.empty(H * W * 4, dtype=torch.float32, device="cuda")
for cname, cfg in (("flat", Config(integrator=IntegratorStrategy.Flat)), ("pt1", Config(max_depth=1))):
    res = {}
    for rnd in range(30):
        for name, a, b in (("rgba8", o8.data_ptr(), 0), ("rgba32f", 0, o32.data_ptr()), ("both", o8.data_ptr(), o32.data_ptr())):
            st = ds.render_device(cfg, W, H, a, b, want_stats=True)
            if rnd >= 10: res.setdefault(name, []).append(st["kernel_ms"])
    print(cname, " | ".join(f"{n} med {np.median(v):.3f} min {min(v):.3f}" for n, v in res.items()), flush=True)
    # host-output path: the same frame through rayca_hip_render (download included by the caller, not by kernel_ms)
    v = [ds.render(cfg, W, H, want_f32=False)[2]["kernel_ms"] for _ in range(10)][3:]
    w = [ds.render(cfg, W, H, want_f32=True, want_rgba8=False)[2]["kernel_ms"] for _ in range(10)][3:]
    print(cname, f"host rgba8 med {np.median(v):.3f} | host rgba32f med {np.median(w):.3f}", flush=True)
