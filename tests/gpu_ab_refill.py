"""A/B of lane-refill kernel variants on Flat frames, F frames in flight, rounds interleaved; prints ms/frame and the lane
utilisation of the node / leaf loops from a counting run.
usage: python tests/gpu_ab_refill.py <soup|atrium> <F> <format 0..3> <variant> [...]   ('main' = shipped .so; 'main:gen' = its generation kernel)"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RAYCA_NODE_FORMAT"] = sys.argv[3]
import numpy as np
import torch
from rayca_amd import Config, DeviceScene, IntegratorStrategy, flatten, scenes, abi
wl, F, names = sys.argv[1], int(sys.argv[2]), sys.argv[4:]
vdir = os.path.join(ROOT, "rayca_amd", "csrc", "variants")
desc, W, H = (flatten(scenes.atrium_scene()), 1920, 1080) if wl == "atrium" else (flatten(scenes.soup_scene()), 4096, 4096)
cfg = Config(integrator=IntegratorStrategy.Flat)
dev = torch.device("cuda", 0)
streams = [torch.cuda.Stream(dev) for _ in range(F)]
outs = [torch.empty((H, W, 4), dtype=torch.uint8, device=dev) for _ in range(F)]
dss, cam = {}, {}
for n in names:
    base, _, kind = n.partition(":")
    path = os.path.join(ROOT, "rayca_amd", "csrc", "librayca_hip.so") if base == "main" else os.path.join(vdir, f"librayca_{base}.so")
    dss[n] = DeviceScene(desc, Config(), builder=abi.BUILDER_SAH, _lib=abi.bind_product_signatures(C.CDLL(path)))
    cam[n] = abi.CAMERA_GENERATION if kind == "gen" else abi.CAMERA_REFILL
res = {n: [] for n in names}
K = 40 if wl == "atrium" else 10
ref = None
for rnd in range(5):
    for n in names:
        ds = dss[n]
        for i in range(F):
            ds.render_device(cfg, W, H, outs[i].data_ptr(), 0, stream=streams[i].cuda_stream, context=i, camera_rays=cam[n])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            i = k % F
            ds.render_device(cfg, W, H, outs[i].data_ptr(), 0, stream=streams[i].cuda_stream, context=i, camera_rays=cam[n])
        torch.cuda.synchronize()
        if rnd: res[n].append((time.perf_counter() - t0) / K * 1e3)
        if rnd == 0:
            frame = outs[0].cpu().numpy()
            if ref is None: ref = frame
            assert np.array_equal(frame, ref), f"{n}: frame differs"
for n in names:
    st = dss[n].render_device(cfg, W, H, outs[0].data_ptr(), 0, stream=streams[0].cuda_stream, want_stats=True, collect_stats=True, camera_rays=cam[n])
    print(f"{n:16s} med {np.median(res[n]):.4f} min {min(res[n]):.4f} ms | node util {st['boxes_tested'] / max(st['wave_box_slots'], 1):.3f} leaf util "
          f"{st['triangles_tested'] / max(st['wave_triangle_slots'], 1):.3f}", flush=True)
