"""Where a frame's time goes, by kernel: the wavefront engine runs trace / shade / shadow as separate kernels.
usage (under rocprofv3 --kernel-trace --stats): python tests/gpu_wf_split.py [workload]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rayca_amd import Config, DeviceScene, flatten, scenes, abi
wl = sys.argv[1] if len(sys.argv) > 1 else "atrium"
if wl == "atrium": desc = flatten(scenes.atrium_scene()); W, H = 1920, 1080
elif wl == "soup": desc = flatten(scenes.soup_scene()); W, H = 2048, 2048
else: desc = flatten(scenes.cornell_scene()); W, H = 1920, 1080
ds = DeviceScene(desc, Config(), builder=abi.BUILDER_SAH)
for i in range(12):
    st = ds.render(Config(max_depth=1), W, H, want_f32=False, engine=abi.ENGINE_WAVEFRONT)[2]
print(wl, "wavefront pt1 kernel_ms", st["kernel_ms"])
