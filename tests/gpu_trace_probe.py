"""Lean trace-only kernel against the fused generation kernel on the same primary rays (atrium 1080p, 8x8 tile order).
usage: python tests/gpu_trace_probe.py <variant> [...]"""
import ctypes as C, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rayca_amd import Config, DeviceScene, IntegratorStrategy, flatten, scenes, abi
from rayca_amd.model import quat_axis_angle
W, H = 1920, 1080
# camera rays as scene.rs:125-141 builds them (values need not be bit-identical for a timing probe)
ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
ang = np.float32(math.tan(math.radians(60.0) / 2)); asp = np.float32(W / H)
xx = (2 * ((xs + 0.5) / W) - 1) * ang * asp
yy = (1 - 2 * ((ys + 0.5) / H)) * ang
d = np.stack([xx, yy, -np.ones_like(xx)], -1)
d /= np.linalg.norm(d, axis=-1, keepdims=True)
# rotate -90 deg about Y: (x, y, z) -> (-z... ) use the quaternion
q = np.array(quat_axis_angle((0.0, 1.0, 0.0), -math.pi / 2), np.float32)
u = q[:3]; s = q[3]
d = 2 * (d @ u)[..., None] * u + (s * s - u @ u) * d + 2 * s * np.cross(u, d)
o = np.broadcast_to(np.array([-15.0 + 1.0, 2.2, 0.3], np.float32), d.shape)
rays = np.concatenate([o, d], -1).astype(np.float32)            # (H, W, 6)
tiles = rays.reshape(H // 8, 8, W // 8, 8, 6).transpose(0, 2, 1, 3, 4).reshape(-1, 6)  # 8x8 tile order
rays_d = None
vdir = os.path.join(ROOT, "rayca_amd", "csrc", "variants")
desc = flatten(scenes.atrium_scene())
res = {}
scn = {}
for n in sys.argv[1:]:
    path = os.path.join(ROOT, "rayca_amd", "csrc", "librayca_hip.so") if n == "main" else os.path.join(vdir, f"librayca_{n}.so")
    scn[n] = DeviceScene(desc, Config(), builder=abi.BUILDER_SAH, _lib=abi.bind_product_signatures(C.CDLL(path)))
    res[n] = []
flat = []
for rnd in range(6):
    for n, ds in scn.items():
        t, prim, uv, st = ds.trace_rays(tiles)
        if rnd: res[n].append(st["trace_kernel_ms"])
    st = scn[sys.argv[1]].render(Config(integrator=IntegratorStrategy.Flat), W, H, want_f32=False)[2]
    if rnd: flat.append(st["kernel_ms"])
hits = int((prim != 0xFFFFFFFF).sum())
print("hits", hits, "of", tiles.shape[0])
print("flat k_generation", np.median(flat))
for n, v in res.items():
    print(f"{n:10s} k_trace_rays med {np.median(v):.4f} min {min(v):.4f} ms")
