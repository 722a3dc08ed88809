"""How fast is the per-pixel stack machine?  Atrium 1080p, a few Configs only it can render, and the common
Pathtracer config on both engines for scale."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rayca_amd import Config, DeviceScene, IntegratorStrategy as I, SamplerStrategy as S, flatten, scenes, abi
ds = DeviceScene(flatten(scenes.atrium_scene()), Config(), builder=abi.BUILDER_SAH)
W, H = 1920, 1080
cases = [("pathtracer depth 3 (generation kernels)", Config(max_depth=3), abi.ENGINE_AUTO),
         ("pathtracer depth 3 (stack machine)", Config(max_depth=3), abi.ENGINE_GENERAL),
         ("raytracer depth 2", Config(integrator=I.Raytracer, max_depth=2), abi.ENGINE_AUTO),
         ("scratcher depth 2", Config(integrator=I.Scratcher, max_depth=2), abi.ENGINE_AUTO),
         ("analytic direct", Config(integrator=I.AnalyticDirect), abi.ENGINE_AUTO),
         ("pathtracer depth 3, 4 light samples", Config(max_depth=3, light_samples=4), abi.ENGINE_AUTO),
         ("pathtracer roulette", Config(russian_roulette=True), abi.ENGINE_AUTO)]
for name, cfg, eng in cases:
    v = []
    for r in range(4):
        st = ds.render(cfg, W, H, want_f32=False, engine=eng)[2]
        if r: v.append(st["kernel_ms"])
    rays = st["rays_primary"] + st["rays_shadow"] + st["rays_bounce"]
    print(f"{name:45s} {np.median(v):9.3f} ms  {rays / 1e6:8.2f} Mrays  {rays / np.median(v) / 1e3:8.1f} Mrays/s", flush=True)
