import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rayca_amd import Config, DeviceScene, IntegratorStrategy, flatten, scenes, abi
wl = sys.argv[1] if len(sys.argv) > 1 else "atrium"
if wl == "atrium": desc = flatten(scenes.atrium_scene()); W,H=1920,1080
elif wl == "soup": desc = flatten(scenes.soup_scene()); W,H=4096,4096
else: desc = flatten(scenes.cornell_scene()); W,H=1920,1080
ds = DeviceScene(desc, Config(), builder=abi.BUILDER_SAH)
print(ds.info())
for name,cfg in (("flat",Config(integrator=IntegratorStrategy.Flat)),("pt1",Config(max_depth=1))):
    st = ds.render(cfg, W, H, want_f32=False, collect_stats=True)[2]
    st2 = ds.render(cfg, W, H, want_f32=False)[2]
    rays = st["rays_primary"]+st["rays_shadow"]
    print(name, "ms", st2["kernel_ms"], "boxes/ray", st["boxes_tested"]/rays, "tris/ray", st["triangles_tested"]/rays,
          "node-loop lane utilisation", st["boxes_tested"]/max(st["wave_box_slots"],1), "leaf-loop utilisation", st["triangles_tested"]/max(st["wave_triangle_slots"],1))
