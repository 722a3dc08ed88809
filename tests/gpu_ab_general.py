"""A/B of library variants on the per-pixel stack machine (k_general): atrium 1080p, Configs only it renders.
usage: python tests/gpu_ab_general.py <variant> [...]   ('main' = the shipped .so)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rayca_amd import Config, DeviceScene, IntegratorStrategy as I, SamplerStrategy as S, flatten, scenes, abi
names = sys.argv[1:]
vdir = os.path.join(ROOT, "rayca_amd", "csrc", "variants")
desc = flatten(scenes.atrium_scene())
W, H = 1920, 1080
cases = [("pt3 general", Config(max_depth=3), abi.ENGINE_GENERAL), ("raytracer 2", Config(integrator=I.Raytracer, max_depth=2), abi.ENGINE_AUTO),
         ("pt3 ls4", Config(max_depth=3, light_samples=4), abi.ENGINE_AUTO)]
dss = {}
for n in names:
    path = os.path.join(ROOT, "rayca_amd", "csrc", "librayca_hip.so") if n == "main" else os.path.join(vdir, f"librayca_{n}.so")
    dss[n] = DeviceScene(desc, Config(), builder=abi.BUILDER_SAH, _lib=abi.bind_product_signatures(C.CDLL(path)))
res = {n: {c[0]: [] for c in cases} for n in names}
ref = {}
for rnd in range(4):
    for cname, cfg, eng in cases:
        for n in names:
            u8, _, st = dss[n].render(cfg, W, H, want_f32=False, engine=eng)
            if rnd: res[n][cname].append(st["kernel_ms"])
            elif cname not in ref: ref[cname] = u8
            else: assert np.array_equal(u8, ref[cname]), (n, cname)
for n in names:
    print(f"{n:10s}", " | ".join(f"{c} {np.median(v):.2f} ms" for c, v in res[n].items()), flush=True)
