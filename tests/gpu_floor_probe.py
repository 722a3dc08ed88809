"""Fixed cost of one frame: the same scene rendered at shrinking sizes (full tile = whole image)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rayca_amd import Config, DeviceScene, IntegratorStrategy, flatten, scenes, abi
ds = DeviceScene(flatten(scenes.atrium_scene()), Config(), builder=abi.BUILDER_SAH)
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(dev)
for name, cfg in (("pt1", Config(max_depth=1)), ("flat", Config(integrator=IntegratorStrategy.Flat))):
    for (W, H) in ((1920, 1080), (960, 540), (480, 270), (240, 135), (64, 64), (8, 8)):
        out = torch.empty((H, W, 4), dtype=torch.uint8, device=dev)
        for _ in range(10):
            ds.render_device(cfg, W, H, out.data_ptr(), 0, stream=stream.cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 200
        for _ in range(K):
            ds.render_device(cfg, W, H, out.data_ptr(), 0, stream=stream.cuda_stream)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / K * 1e3
        st = ds.render_device(cfg, W, H, out.data_ptr(), 0, stream=stream.cuda_stream, want_stats=True)
        print(f"{name} {W}x{H}: {t:.4f} ms/step, kernels {st['kernel_ms']:.4f} ms, trace {st['trace_kernel_ms']:.4f}", flush=True)
