"""A library variant must render the main library's bits: python tests/gpu_variant_equal.py <variant> [...]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rayca_amd import Config, DeviceScene, IntegratorStrategy, flatten, scenes, abi
vdir = os.path.join(ROOT, "rayca_amd", "csrc", "variants")
cases = [("atrium", scenes.atrium_scene, 960, 540, [Config(integrator=IntegratorStrategy.Flat), Config(max_depth=1), Config(max_depth=3, seed=2)]),
         ("soup64k", lambda: scenes.soup_scene(1 << 16, extent=0.04), 512, 512, [Config(integrator=IntegratorStrategy.Flat)]),
         ("cornell", scenes.cornell_scene, 640, 360, [Config(max_depth=4, seed=3)])]
ok = True
for name, mk, w, h, cfgs in cases:
    desc = flatten(mk())
    main = DeviceScene(desc, Config(), builder=abi.BUILDER_SAH)
    main.finish()
    for v in sys.argv[1:]:
        lib = abi.bind_product_signatures(C.CDLL(os.path.join(vdir, f"librayca_{v}.so")))
        other = DeviceScene(desc, Config(), builder=abi.BUILDER_SAH, _lib=lib)
        other.finish()
        for cfg in cfgs:
            for fmt in ("0", None):   # binary f32 nodes pinned is a process-wide env: only the scene's own choice is compared here
                a = main.render(cfg, w, h, want_rgba8=False)[1]
                b = other.render(cfg, w, h, want_rgba8=False)[1]
                same = np.array_equal(a.view(np.uint32), b.view(np.uint32))
                ok &= same
                print(f"{name} {v} depth {cfg.max_depth} integrator {cfg.integrator}: {'same bits' if same else 'DIFFERENT'}", flush=True)
                break
        other.close()
    main.close()
sys.exit(0 if ok else 1)
