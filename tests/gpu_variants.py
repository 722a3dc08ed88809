"""Ad-hoc: time library variants (tests/build_variants.sh) on the bench workload, each in its own process."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vdir = os.path.join(ROOT, "rayca_amd", "csrc", "variants")
names = sys.argv[2:] or sorted(f[len("librayca_"):-3] for f in os.listdir(vdir) if f.endswith(".so"))
wl = sys.argv[1] if len(sys.argv) > 1 else "atrium"
code = r'''
import sys, os, time
sys.path.insert(0, %r)
import numpy as np
from rayca_amd import Config, DeviceScene, IntegratorStrategy, flatten, scenes, abi
wl = %r
if wl == "atrium": desc = flatten(scenes.atrium_scene()); W,H=1920,1080; cfgs=[("pt1",Config(max_depth=1)),("flat",Config(integrator=IntegratorStrategy.Flat)),("pt5",Config())]
elif wl == "soup": desc = flatten(scenes.soup_scene()); W,H=4096,4096; cfgs=[("flat",Config(integrator=IntegratorStrategy.Flat))]
else: desc = flatten(scenes.cornell_scene()); W,H=1920,1080; cfgs=[("flat",Config(integrator=IntegratorStrategy.Flat)),("pt1",Config(max_depth=1))]
ds = DeviceScene(desc, Config(), builder=abi.BUILDER_SAH)
out=[]
for name,cfg in cfgs:
    st = ds.render(cfg, W, H, want_f32=False)[2]
    ms=[]
    for i in range(5):
        st = ds.render(cfg, W, H, want_f32=False)[2]; ms.append(st["kernel_ms"])
    rays = st["rays_primary"]+st["rays_shadow"]+st["rays_bounce"]
    out.append("%%s %%.3f ms %%.0f Mrays/s" %% (name, min(ms), rays/min(ms)/1e3))
print(" | ".join(out))
''' % (ROOT, wl)
for n in names:
    env = dict(os.environ, RAYCA_HIP_LIB=os.path.join(vdir, f"librayca_{n}.so"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=280)
    print(f"{n:12s}", r.stdout.strip() or r.stderr.strip()[-300:], flush=True)
