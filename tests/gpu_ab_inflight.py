"""A/B of library variants with frames in flight (the bench's operating point): each variant renders K frames on F
frame contexts/streams, rounds interleaved.  usage: python tests/gpu_ab_inflight.py <workload> <F> <variant> [...]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from rayca_amd import Config, DeviceScene, IntegratorStrategy, flatten, scenes, abi
wl, F, names = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
vdir = os.path.join(ROOT, "rayca_amd", "csrc", "variants")
if wl == "atrium": desc = flatten(scenes.atrium_scene()); W, H = 1920, 1080; cfgs = [("pt1", Config(max_depth=1)), ("flat", Config(integrator=IntegratorStrategy.Flat)), ("pt5", Config())]
elif wl == "atrium4k": desc = flatten(scenes.atrium_scene()); W, H = 3840, 2160; cfgs = [("pt4", Config(max_depth=4))]
else: desc = flatten(scenes.soup_scene()); W, H = 4096, 4096; cfgs = [("flat", Config(integrator=IntegratorStrategy.Flat))]
dev = torch.device("cuda", 0)
from rayca_amd.streams import frame_streams
streams = frame_streams(dev, F, spare=0)[0]   # on different hardware queues (rayca_amd/streams.py)
outs = [torch.empty((H, W, 4), dtype=torch.uint8, device=dev) for _ in range(F)]
dss = {}
for n in names:
    path = os.path.join(ROOT, "rayca_amd", "csrc", "librayca_hip.so") if n == "main" else os.path.join(vdir, f"librayca_{n}.so")
    dss[n] = DeviceScene(desc, Config(), builder=abi.BUILDER_SAH, _lib=abi.bind_product_signatures(C.CDLL(path)))
res = {n: {c: [] for c, _ in cfgs} for n in names}
K = 60 if wl == "atrium" else 12
for rnd in range(6):
    for cname, cfg in cfgs:
        for n in names:
            ds = dss[n]
            for i in range(F):   # contexts warm, node format decided
                for _ in range(2 if rnd else 12):
                    ds.render_device(cfg, W, H, outs[i].data_ptr(), 0, stream=streams[i].cuda_stream, context=i, want_stats=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(K):
                i = k % F
                ds.render_device(cfg, W, H, outs[i].data_ptr(), 0, stream=streams[i].cuda_stream, context=i)
            torch.cuda.synchronize()
            if rnd: res[n][cname].append((time.perf_counter() - t0) / K * 1e3)
for n in names:
    print(f"{n:12s}", " | ".join(f"{c} med {np.median(v):.4f} min {min(v):.4f} ms" for c, v in res[n].items()), flush=True)
