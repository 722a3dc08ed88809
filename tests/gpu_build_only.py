"""scene_create a few times and nothing else (for rocprofv3 --kernel-trace --stats of the builder): python tests/gpu_build_only.py [atrium|soup] [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rayca_amd import Config, DeviceScene, flatten, scenes, abi
wl = sys.argv[1] if len(sys.argv) > 1 else "atrium"
desc = flatten(scenes.atrium_scene() if wl == "atrium" else scenes.soup_scene())
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
    ds = DeviceScene(desc, Config(), builder=abi.BUILDER_SAH)
    ds.finish()
    print(ds.info()["build_ms"], flush=True)
    ds.close()
