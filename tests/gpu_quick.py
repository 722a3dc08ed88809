"""Ad-hoc GPU shake-out (not a pytest): renders small scenes through the C ABI and compares with the oracle."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol
from rayca_amd import Config, DeviceScene, IntegratorStrategy, flatten, scenes, abi

def cmp(name, ds, orc, cfg, w, h, **kw):
    t = time.time(); u8, f32, st = ds.render(cfg, w, h, **kw); tg = time.time() - t
    t = time.time(); ou8, of32, ost = orc.render(cfg, w, h); tc = time.time() - t
    d = np.abs(f32 - of32)
    nbad = int((d.max(-1) > 1e-4).sum())
    print(f"{name}: max|d|={d.max():.3e} bad_px={nbad}/{w*h} exact={np.array_equal(f32, of32)} u8diff={int(np.abs(u8.astype(int)-ou8.astype(int)).max())} "
          f"gpu {tg*1e3:.1f} ms (kernel {st['kernel_ms']:.3f} ms, trace {st['trace_kernel_ms']:.3f}) cpu {tc*1e3:.1f} ms", flush=True)
    print("   gpu stats", {k: st[k] for k in ('rays_primary','rays_shadow','rays_bounce','boxes_tested','triangles_tested','hits_shaded')}, flush=True)
    print("   cpu stats", {k: ost[k] for k in ('rays_primary','rays_shadow','rays_bounce','boxes_tested','triangles_tested','hits_shaded')}, flush=True)
    return u8, f32, of32

print("devices", __import__('rayca_amd.lib', fromlist=['x']).load().rayca_hip_device_count(), flush=True)
desc = flatten(scenes.box_scene())
ds = DeviceScene(desc, Config()); print(ds.info(), flush=True)
orc = ol.OracleScene(desc, Config())
print("prim order equal:", np.array_equal(ds.primitive_order(), orc.primitive_order()), flush=True)
cmp("box flat", ds, orc, Config(integrator=IntegratorStrategy.Flat), 256, 256, collect_stats=True)
cmp("box flat exhaustive", ds, orc, Config(integrator=IntegratorStrategy.Flat), 256, 256, collect_stats=True, traversal=abi.TRAVERSAL_EXHAUSTIVE)
cmp("box pt md1", ds, orc, Config(max_depth=1), 256, 256, collect_stats=True)
cmp("box pt md5", ds, orc, Config(), 256, 256, collect_stats=True)
cmp("box pt md5 spp4", ds, orc, Config(samples_per_pixel=4), 128, 128, collect_stats=True)
desc = flatten(scenes.cornell_scene())
ds = DeviceScene(desc, Config()); print(ds.info(), flush=True)
orc = ol.OracleScene(desc, Config())
cmp("cornell flat", ds, orc, Config(integrator=IntegratorStrategy.Flat), 640, 360, collect_stats=True)
cmp("cornell pt md1", ds, orc, Config(max_depth=1), 640, 360, collect_stats=True)
u8, f32, of32 = cmp("cornell pt md5", ds, orc, Config(), 640, 360, collect_stats=True)
from pngdump import write_png
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
write_png(os.path.join(ROOT, "gpurun_out", "cornell_pt5_gpu.png"), u8)
desc = flatten(scenes.soup_scene(20000))
ds = DeviceScene(desc, Config()); print(ds.info(), flush=True)
orc = ol.OracleScene(desc, Config(), build=ol.BUILD_BINNED)
print("prim order equal:", np.array_equal(ds.primitive_order(), orc.primitive_order()), flush=True)
u8, f32, of32 = cmp("soup20k flat", ds, orc, Config(integrator=IntegratorStrategy.Flat), 512, 512, collect_stats=True)
cmp("soup20k flat exhaustive", ds, orc, Config(integrator=IntegratorStrategy.Flat), 512, 512, collect_stats=True, traversal=abi.TRAVERSAL_EXHAUSTIVE)
write_png(os.path.join(ROOT, "gpurun_out", "soup20k_gpu.png"), u8)
