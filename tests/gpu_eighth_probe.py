"""Where an eighth of the 1080p frame (one rank's share of an 8-GPU run) spends its time on this GPU: kernel classes of one
frame alone, wall clock of one frame alone, and frames per second with F = 1..8 frame contexts in flight (no exchange).
usage: python tests/gpu_eighth_probe.py [parts ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from rayca_amd import DeviceScene, flatten, scenes, abi
from rayca_amd.streams import frame_streams
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
w = bench.workload_config("atrium")
cfg, W, H = w["cfg"], w["width"], w["height"]
fs, sp = frame_streams(dev, 8)
streams = fs + sp
ds = DeviceScene(flatten(scenes.atrium_scene()), cfg, builder=abi.BUILDER_SAH)
ds.finish()
for parts in [int(a) for a in sys.argv[1:]] or [1, 8]:
    tile = (0, parts, 8)
    rows = ds.tile_rows(tile, H)
    outs = [torch.empty((rows, W, 4), dtype=torch.uint8, device=dev) for _ in range(8)]
    for i in range(8):
        for _ in range(12):
            ds.render_device(cfg, W, H, outs[i].data_ptr(), 0, tile=tile, stream=streams[i].cuda_stream, context=i, want_stats=True)
    torch.cuda.synchronize()
    cm = np.zeros(8); cn = np.zeros(8); km = []
    for _ in range(20):
        st = ds.render_device(cfg, W, H, outs[0].data_ptr(), 0, tile=tile, stream=streams[0].cuda_stream, context=0, want_stats=True)
        cm += np.array(st["class_ms"]); cn += np.array(st["class_launches"]); km.append(st["kernel_ms"])
    print(f"parts {parts} ({rows} rows): kernels of one frame {np.mean(km):.4f} ms: " + ", ".join(f"{abi.KERNEL_NAMES[k]} {cm[k] / 20:.4f} ({cn[k] / 20:.0f})" for k in range(8) if cn[k] > 0), flush=True)
    issue = [ds.prepare_device(cfg, W, H, outs[i].data_ptr(), 0, tile=tile, stream=streams[i].cuda_stream, context=i) for i in range(8)]
    for _ in range(5):
        issue[0](); streams[0].synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        issue[0](); streams[0].synchronize()
    print(f"   one frame at a time, issue + synchronise: {(time.perf_counter() - t0) / 100 * 1e3:.4f} ms", flush=True)
    for F in (1, 2, 3, 4, 6, 8):
        K = 400
        for k in range(2 * F):
            issue[k % F]()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            issue[k % F]()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"   F={F}: {(t2 - t0) / K * 1e3:.4f} ms/frame (host issue {(t1 - t0) / K * 1e6:.1f} us/frame)", flush=True)
ds.close()
