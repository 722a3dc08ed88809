"""scene_create timing breakdown (RAYCA_BUILD_TIMING=1 prints the laps to stderr).  usage: python tests/gpu_build_probe.py [atrium|soup] [n]"""
import os, sys, time
os.environ["RAYCA_BUILD_TIMING"] = "1"
os.environ["RAYCA_RENDER_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rayca_amd import Config, DeviceScene, flatten, scenes, abi
wl = sys.argv[1] if len(sys.argv) > 1 else "atrium"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
desc = flatten(scenes.atrium_scene() if wl == "atrium" else scenes.soup_scene())
for i in range(n):
    t0 = time.perf_counter()
    ds = DeviceScene(desc, Config(), builder=abi.BUILDER_SAH)
    t1 = time.perf_counter()
    print(f"[probe] {wl} scene_create #{i}: wall {1e3 * (t1 - t0):.1f} ms, build_ms {ds.info()['build_ms']:.1f}", file=sys.stderr, flush=True)
    # the literal drop-in draw(): build, one frame, destroy -- the frame runs on the binary nodes while the other formats are made
    u8, _, st = ds.render(Config(max_depth=1), 1920, 1080, want_f32=False)
    t2 = time.perf_counter()
    ds.finish()
    t3 = time.perf_counter()
    ds.close()
    t4 = time.perf_counter()
    print(f"[probe]   first frame +{1e3 * (t2 - t1):.1f} ms (node_format {st['node_format']}), other formats ready +{1e3 * (t3 - t2):.1f} ms, destroy {1e3 * (t4 - t3):.1f} ms",
          file=sys.stderr, flush=True)
