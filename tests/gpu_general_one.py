"""One Config on the stack machine (k_general), a few frames: for rocprofv3.  usage: python tests/gpu_general_one.py [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rayca_amd import Config, DeviceScene, flatten, scenes, abi
ds = DeviceScene(flatten(scenes.atrium_scene()), Config(), builder=abi.BUILDER_SAH)
ds.finish()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    st = ds.render(Config(max_depth=3), 1920, 1080, want_f32=False, engine=abi.ENGINE_GENERAL)[2]
print(st["kernel_ms"])
ds.close()
