"""k_general A/B of library variants on Pathtracer depth 3 (atrium 1080p, stack machine): python tests/gpu_general_ab.py <variant> ..."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rayca_amd import Config, DeviceScene, flatten, scenes, abi
desc = flatten(scenes.atrium_scene())
vdir = os.path.join(ROOT, "rayca_amd", "csrc", "variants")
ref = None
for n in ["main"] + sys.argv[1:]:
    path = os.path.join(ROOT, "rayca_amd", "csrc", "librayca_hip.so") if n == "main" else os.path.join(vdir, f"librayca_{n}.so")
    ds = DeviceScene(desc, Config(), builder=abi.BUILDER_SAH, _lib=abi.bind_product_signatures(C.CDLL(path)))
    ds.finish()
    for label, cfg in (("depth 3", Config(max_depth=3)), ("depth 3, 4 light samples", Config(max_depth=3, light_samples=4))):
        v = []
        for r in range(4):
            u8, f32, st = ds.render(cfg, 1920, 1080, engine=abi.ENGINE_GENERAL)
            if r: v.append(st["kernel_ms"])
        same = ""
        if label == "depth 3":
            if ref is None: ref = f32.copy()
            else: same = "same bits" if np.array_equal(ref.view(np.uint32), f32.view(np.uint32)) else "DIFFERENT"
        print(f"{n:10s} {label:26s} {np.median(v):9.3f} ms {same}", flush=True)
    ds.close()
