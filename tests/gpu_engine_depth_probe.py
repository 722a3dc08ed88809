"""Which engine for how many generations?  Fused and wavefront engines on path frames of depth 1..5, four frames in
flight (the bench's operating point).  usage: python tests/gpu_engine_depth_probe.py [workload]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rayca_amd import Config, DeviceScene, flatten, scenes, abi
wl = sys.argv[1] if len(sys.argv) > 1 else "atrium"
desc = flatten(scenes.atrium_scene() if wl == "atrium" else scenes.cornell_scene())
W, H, F = 1920, 1080, 4
dev = torch.device("cuda", 0)
from rayca_amd.streams import frame_streams
streams = frame_streams(dev, F, spare=0)[0]   # each on a hardware queue of its own
ds = DeviceScene(desc, Config(), builder=abi.BUILDER_SAH)
ds.finish()
outs = [torch.empty((H, W, 4), dtype=torch.uint8, device=dev) for _ in range(F)]
for depth in (1, 2, 3, 4, 5):
    cfg = Config(max_depth=depth)
    row = []
    for ename, eng in (("fused", abi.ENGINE_FUSED), ("wavefront", abi.ENGINE_WAVEFRONT)):
        for i in range(F):
            for _ in range(10):   # contexts warm, node formats decided
                ds.render_device(cfg, W, H, outs[i].data_ptr(), 0, stream=streams[i].cuda_stream, context=i, want_stats=True, engine=eng)
        best = 1e9
        for rnd in range(3):
            torch.cuda.synchronize()
            K = 40
            t0 = time.perf_counter()
            for k in range(K):
                ds.render_device(cfg, W, H, outs[k % F].data_ptr(), 0, stream=streams[k % F].cuda_stream, context=k % F, engine=eng)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / K * 1e3)
        row.append(f"{ename} {best:.3f} ms")
    print(f"{wl} depth {depth}: " + " | ".join(row), flush=True)
