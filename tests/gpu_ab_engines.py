"""Fused vs wavefront engine, interleaved in one process.  usage: python tests/gpu_ab_engines.py [atrium|soup|cornell]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rayca_amd import Config, DeviceScene, IntegratorStrategy, flatten, scenes, abi
wl = sys.argv[1] if len(sys.argv) > 1 else "atrium"
if wl == "atrium": desc = flatten(scenes.atrium_scene()); W, H = 1920, 1080; cfgs = [("pt1", Config(max_depth=1)), ("flat", Config(integrator=IntegratorStrategy.Flat)), ("pt5", Config())]
elif wl == "soup": desc = flatten(scenes.soup_scene()); W, H = 4096, 4096; cfgs = [("flat", Config(integrator=IntegratorStrategy.Flat))]
else: desc = flatten(scenes.cornell_scene()); W, H = 1920, 1080; cfgs = [("flat", Config(integrator=IntegratorStrategy.Flat)), ("pt1", Config(max_depth=1)), ("pt5", Config())]
ds = DeviceScene(desc, Config(), builder=abi.BUILDER_SAH)
engines = {"fused": abi.ENGINE_FUSED, "wavefront": abi.ENGINE_WAVEFRONT}
res = {e: {c: [] for c, _ in cfgs} for e in engines}
tr = {e: {c: [] for c, _ in cfgs} for e in engines}
for rnd in range(7):
    for cname, cfg in cfgs:
        for e, ev in engines.items():
            st = ds.render(cfg, W, H, want_f32=False, engine=ev)[2]
            if rnd:
                res[e][cname].append(st["kernel_ms"]); tr[e][cname].append(st["trace_kernel_ms"])
for e in engines:
    print(f"{e:10s}", " | ".join(f"{c} med {np.median(v):.3f} (trace {np.median(tr[e][c]):.3f}) ms" for c, v in res[e].items()), flush=True)
