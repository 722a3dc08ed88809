"""Fused vs wavefront engine across scene sizes (atrium at several tessellation levels)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rayca_amd import Config, DeviceScene, IntegratorStrategy, flatten, scenes, abi
W, H = 1920, 1080
cfgs = [("pt1", Config(max_depth=1)), ("flat", Config(integrator=IntegratorStrategy.Flat)), ("pt3", Config(max_depth=3))]
for detail in (2, 4, 6, 9):
    ds = DeviceScene(flatten(scenes.atrium_scene(detail)), Config(), builder=abi.BUILDER_SAH)
    info = ds.info()
    out = []
    for cname, cfg in cfgs:
        r = {}
        for e, ev in (("fused", abi.ENGINE_FUSED), ("wf", abi.ENGINE_WAVEFRONT)):
            v = []
            for rnd in range(5):
                st = ds.render(cfg, W, H, want_f32=False, engine=ev)[2]
                if rnd: v.append(st["kernel_ms"])
            r[e] = np.median(v)
        out.append(f"{cname} fused {r['fused']:.3f} wf {r['wf']:.3f}")
    print(f"detail {detail}: {info['triangle_count']} tris, device {info['device_bytes'] / 1e6:.1f} MB | " + " | ".join(out), flush=True)
