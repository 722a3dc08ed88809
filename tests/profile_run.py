"""Workload for rocprofv3 passes: the bench configuration (atrium 1080p, primary + 1 shadow), a few frames.
usage: python3 tests/profile_run.py [atrium|soup] [pt1|flat] [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rayca_amd import Config, DeviceScene, IntegratorStrategy, flatten, scenes, abi
wl = sys.argv[1] if len(sys.argv) > 1 else "atrium"
mode = sys.argv[2] if len(sys.argv) > 2 else "pt1"
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 5
desc, W, H = (flatten(scenes.atrium_scene()), 1920, 1080) if wl == "atrium" else (flatten(scenes.soup_scene()), 4096, 4096)
cfg = Config(max_depth=1) if mode == "pt1" else Config(integrator=IntegratorStrategy.Flat)
ds = DeviceScene(desc, cfg, builder=abi.BUILDER_SAH)
for _ in range(frames):
    st = ds.render(cfg, W, H, want_f32=False)[2]
print(wl, mode, st["kernel_ms"], st["trace_kernel_ms"])
