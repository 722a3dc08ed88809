"""Would sorting a bounce generation's rays pay?  Bounce-like rays of the atrium (origins = hit points of a tiled camera
grid, directions random over the hemisphere facing back) traced by k_trace_rays in queue order (tile by tile, as the
compaction writes them), and sorted by direction octant, by origin cell (Morton, 2^-k of the scene), and by both.
usage: python tests/gpu_sort_probe.py   (RAYCA_NODE_FORMAT=0 binary / 1 4-wide nodes)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rayca_amd import Config, DeviceScene, flatten, scenes, abi
ds = DeviceScene(flatten(scenes.atrium_scene()), Config(), builder=abi.BUILDER_SAH)
ds.finish()
W, H = 1920, 1080
rs = np.random.RandomState(5)
# camera-like grid, 8x8 tiles in row-major tile order
ty, tx = np.meshgrid(np.arange(H // 8), np.arange(W // 8), indexing="ij")
py = (ty[..., None, None] * 8 + np.arange(8)[None, None, :, None]).repeat(8, 3).reshape(-1)
px = (tx[..., None, None] * 8 + np.arange(8)[None, None, None, :]).repeat(8, 2).reshape(-1)
u = (px + 0.5) / W * 2 - 1
v = 1 - (py + 0.5) / H * 2
d = np.stack([u * (W / H) * 0.6, v * 0.6, -np.ones_like(u)], 1).astype(np.float32)
d /= np.linalg.norm(d, axis=1, keepdims=True)
o = np.tile(np.array([[0.0, 1.5, 4.0]], np.float32), (d.shape[0], 1))
t, prim, _, st = ds.trace_rays(np.concatenate([o, d], 1))
hit = prim != 0xFFFFFFFF
print(f"camera rays: {hit.mean():.3f} hit, kernel {st['kernel_ms']:.3f} ms", flush=True)
o2 = (o + d * (t[:, None] - 1e-3))[hit].astype(np.float32)
d2 = rs.normal(size=(o2.shape[0], 3)).astype(np.float32)
d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
flip = (d2 * d[hit]).sum(1) > 0
d2[flip] *= -1
rays = np.concatenate([o2, d2], 1)

def timed(order, label):
    r = rays if order is None else rays[order]
    ms = []
    for _ in range(5):
        _, _, _, s = ds.trace_rays(r)
        ms.append(s["kernel_ms"])
    print(f"{label:46s} {np.median(ms):.3f} ms", flush=True)

octant = (d2[:, 0] > 0) * 1 + (d2[:, 1] > 0) * 2 + (d2[:, 2] > 0) * 4
lo, hi = o2.min(0), o2.max(0)
def morton(bits):
    q = np.minimum(((o2 - lo) / (hi - lo + 1e-9) * (1 << bits)).astype(np.int64), (1 << bits) - 1)
    m = np.zeros(len(q), np.int64)
    for b in range(bits):
        for a in range(3):
            m |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return m
timed(None, "queue order (tile by tile)")
timed(rs.permutation(len(rays)), "random order")
timed(np.argsort(octant, kind="stable"), "by direction octant")
for bits in (2, 3, 4, 6):
    m = morton(bits)
    timed(np.argsort(m, kind="stable"), f"by origin cell ({bits} bits/axis)")
    timed(np.lexsort((octant, m)), f"by origin cell ({bits} bits/axis), then octant")
    timed(np.lexsort((m, octant)), f"by octant, then origin cell ({bits} bits/axis)")
# octant within blocks of 256 / 1024 / 4096 consecutive queue entries (what a workgroup / a few could do without a global sort)
for blk in (256, 1024, 4096, 16384):
    key = (np.arange(len(rays)) // blk) * 8 + octant
    timed(np.argsort(key, kind="stable"), f"octant within blocks of {blk} queue entries")
ds.close()
