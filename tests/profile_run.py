"""Workload for rocprofv3 --pmc passes: a few frames of one bench workload, one frame at a time (the same launches
bench.py's roofline loop times).  usage: python3 tests/profile_run.py [atrium|soup|cornell|atrium4k] [frames]
The node format (and the soup's camera-ray kernel) is pinned by the caller through RAYCA_NODE_FORMAT / RAYCA_REFILL to what
the un-profiled bench settles on: counter collection perturbs the scene's own timing of the formats."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from rayca_amd import DeviceScene, flatten, scenes, abi
wl = sys.argv[1] if len(sys.argv) > 1 else "atrium"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 5
w = bench.workload_config(wl)
cfg, W, H = w["cfg"], w["width"], w["height"]
desc = flatten(scenes.WORKLOADS["atrium" if wl == "atrium4k" else wl]["scene"]())
ds = DeviceScene(desc, cfg, builder=abi.BUILDER_SAH)
ds.finish()
for _ in range(frames):
    st = ds.render(cfg, W, H, want_f32=False)[2]
print(wl, st["kernel_ms"], st["trace_kernel_ms"], "node_format", st["node_format"], dict(zip(abi.KERNEL_NAMES, zip(st["class_launches"], st["class_ms"]))))
