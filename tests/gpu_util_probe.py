"""Lane utilisation of the traversal loops: boxes_tested / wave_box_slots, triangles_tested / wave_triangle_slots
(slots = 64 lanes x every trip a wave takes through the node loop / the leaf loop).
usage: python tests/gpu_util_probe.py <node format 0..3>      (one process per format: the pin is read once)"""
import os, sys
os.environ["RAYCA_NODE_FORMAT"] = sys.argv[1] if len(sys.argv) > 1 else "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rayca_amd import Config, DeviceScene, IntegratorStrategy, flatten, scenes, abi
for wl in ("atrium", "soup", "cornell"):
    if wl == "atrium": desc = flatten(scenes.atrium_scene()); W, H = 1920, 1080
    elif wl == "soup": desc = flatten(scenes.soup_scene()); W, H = 2048, 2048
    else: desc = flatten(scenes.cornell_scene()); W, H = 1920, 1080
    ds = DeviceScene(desc, Config(), builder=abi.BUILDER_SAH)
    for cname, cfg in (("flat", Config(integrator=IntegratorStrategy.Flat)), ("pt1", Config(max_depth=1))):
        st = ds.render(cfg, W, H, want_f32=False, collect_stats=True)[2]
        b, s, t, ts = st["boxes_tested"], st["wave_box_slots"], st["triangles_tested"], st["wave_triangle_slots"]
        print(f"fmt {os.environ['RAYCA_NODE_FORMAT']} {wl:8s} {cname:5s}: boxes {b/1e6:8.1f}M slots {s/1e6:8.1f}M util {b/max(s,1):.3f} | tris {t/1e6:7.1f}M slots {ts/1e6:8.1f}M util {t/max(ts,1):.3f}", flush=True)
