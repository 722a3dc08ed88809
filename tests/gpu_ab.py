"""A/B timing of library variants inside ONE process, interleaved rounds (device-to-device and
run-to-run variance make cross-process comparisons unreliable).
usage: python tests/gpu_ab.py <workload> <variant> [<variant> ...]   (variants from tests/build_variants.sh; 'main' = the shipped .so)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rayca_amd import Config, DeviceScene, IntegratorStrategy, flatten, scenes, abi
wl = sys.argv[1]
names = sys.argv[2:]
vdir = os.path.join(ROOT, "rayca_amd", "csrc", "variants")
if wl == "atrium": desc = flatten(scenes.atrium_scene()); W, H = 1920, 1080; cfgs = [("pt1", Config(max_depth=1)), ("flat", Config(integrator=IntegratorStrategy.Flat)), ("pt5", Config())]
elif wl == "soup": desc = flatten(scenes.soup_scene()); W, H = 4096, 4096; cfgs = [("flat", Config(integrator=IntegratorStrategy.Flat))]
else: desc = flatten(scenes.cornell_scene()); W, H = 1920, 1080; cfgs = [("flat", Config(integrator=IntegratorStrategy.Flat)), ("pt1", Config(max_depth=1))]
libs = {}
for n in names:
    path = os.path.join(ROOT, "rayca_amd", "csrc", "librayca_hip.so") if n == "main" else os.path.join(vdir, f"librayca_{n}.so")
    libs[n] = abi.bind_product_signatures(C.CDLL(path))
dss = {n: DeviceScene(desc, Config(), builder=abi.BUILDER_SAH, _lib=l) for n, l in libs.items()}
res = {n: {c: [] for c, _ in cfgs} for n in names}
for rnd in range(24):  # the first 8 large frames of a (scene, mode) calibrate the node format
    for cname, cfg in cfgs:
        for n in names:
            st = dss[n].render(cfg, W, H, want_f32=False)[2]
            if rnd >= 9: res[n][cname].append(st["kernel_ms"])
for n in names:
    print(f"{n:14s}", " | ".join(f"{c} med {np.median(v):.3f} min {min(v):.3f} ms" for c, v in res[n].items()), flush=True)
