"""Closed-form GGX terms (kernels.hip, RAYCA_GGX_CLOSED_FORM) against the reference's libm spelling: distance of
both from the oracle on every shaded parity case, and the speed of the bench frames.
usage: bash tests/build_variants.sh ggxlibm "-DRAYCA_GGX_CLOSED_FORM=0" && python tests/gpu_ggx_probe.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from rayca_amd import Config, DeviceScene, IntegratorStrategy, flatten, scenes, abi
import oracle_lib as ol
import test_gpu_general as G
libs = {"closed": abi.bind_product_signatures(C.CDLL(os.path.join(ROOT, "rayca_amd", "csrc", "librayca_hip.so"))),
        "libm": abi.bind_product_signatures(C.CDLL(os.path.join(ROOT, "rayca_amd", "csrc", "variants", "librayca_ggxlibm.so")))}
TOL = 1e-4
for i, (name, cfg, allowed) in enumerate(G.CASES):
    desc = flatten(G.SCENES[name]())
    orc = ol.OracleScene(desc, Config())
    try:
        _, of32, _ = orc.render(cfg, G.W, G.H)
    except Exception as e:
        print(i, name, "oracle:", e); continue
    b = np.nan_to_num(of32, nan=-1.0)
    row = []
    for ln, lib in libs.items():
        ds = DeviceScene(desc, Config(), _lib=lib)
        _, f32, _ = ds.render(cfg, G.W, G.H)
        a = np.nan_to_num(f32, nan=-1.0)
        err = np.abs(a - b) / np.maximum(1.0, np.abs(b))
        row.append(f"{ln}: beyond 1e-4 {(err > TOL).any(-1).mean():.5f} worst {err.max():.2e} mean {err.mean():.2e}")
        ds.close()
    print(f"case {i:2d} {name:12s} allowed {allowed:.3f} | " + " | ".join(row), flush=True)
# the atrium at a size the oracle renders in seconds (PBR materials, point light): the bench frame's shading
desc = flatten(scenes.atrium_scene(detail=3))
orc = ol.OracleScene(desc, Config(), threads=16)
for cfg, label in ((Config(max_depth=1), "depth 1"), (Config(max_depth=3, seed=3), "depth 3")):
    _, of32, _ = orc.render(cfg, 320, 180)
    b = np.nan_to_num(of32, nan=-1.0)
    row = []
    for ln, lib in libs.items():
        ds = DeviceScene(desc, Config(), builder=abi.BUILDER_REFERENCE, _lib=lib)
        _, f32, _ = ds.render(cfg, 320, 180)
        err = np.abs(np.nan_to_num(f32, nan=-1.0) - b) / np.maximum(1.0, np.abs(b))
        row.append(f"{ln}: beyond 1e-4 {(err > TOL).any(-1).mean():.5f} worst {err.max():.2e}")
        ds.close()
    print(f"atrium(detail 3) {label} | " + " | ".join(row), flush=True)
