"""Where a frame's time goes, by kernel: the wavefront engine runs trace / shade / shadow as separate kernels.
usage (under rocprofv3 --kernel-trace): python tests/gpu_wf_split.py [workload] [max_depth]
prints lane utilisation of the whole frame too (counting instantiation)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rayca_amd import Config, DeviceScene, flatten, scenes, abi
wl = sys.argv[1] if len(sys.argv) > 1 else "atrium"
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if wl == "atrium": desc = flatten(scenes.atrium_scene()); W, H = 1920, 1080
elif wl == "soup": desc = flatten(scenes.soup_scene()); W, H = 2048, 2048
else: desc = flatten(scenes.cornell_scene()); W, H = 1920, 1080
ds = DeviceScene(desc, Config(), builder=abi.BUILDER_SAH)
cfg = Config(max_depth=depth)
for i in range(12):
    st = ds.render(cfg, W, H, want_f32=False, engine=abi.ENGINE_WAVEFRONT)[2]
print(wl, "wavefront depth", depth, "kernel_ms", st["kernel_ms"])
st = ds.render(cfg, W, H, want_f32=False, engine=abi.ENGINE_WAVEFRONT, collect_stats=True)[2]
print("rays", st["rays_primary"], st["rays_shadow"], st["rays_bounce"], "node-loop utilisation", st["boxes_tested"] / max(st["wave_box_slots"], 1),
      "leaf-loop utilisation", st["triangles_tested"] / max(st["wave_triangle_slots"], 1))
