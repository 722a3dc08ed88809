"""Ad-hoc: atrium/soup with both builders, timing + equality."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from rayca_amd import Config, DeviceScene, IntegratorStrategy, flatten, scenes, abi
from pngdump import write_png
FLAT = Config(integrator=IntegratorStrategy.Flat)
which = sys.argv[1] if len(sys.argv) > 1 else "atrium"
if which == "atrium":
    t=time.time(); desc = flatten(scenes.atrium_scene()); print("flatten", time.time()-t, flush=True); W,H = 1920,1080; cfgs=[("flat",FLAT),("pt1",Config(max_depth=1))]
else:
    t=time.time(); desc = flatten(scenes.soup_scene()); print("flatten", time.time()-t, flush=True); W,H = 4096,4096; cfgs=[("flat",FLAT)]
imgs = {}
for bname, b in (("sah", abi.BUILDER_SAH), ("reference", abi.BUILDER_REFERENCE)):
    t=time.time(); ds = DeviceScene(desc, Config(), builder=b); print(bname, "build", time.time()-t, ds.info(), flush=True)
    for cname, cfg in cfgs:
        for trav in ((abi.TRAVERSAL_ORDERED,"ordered"),(abi.TRAVERSAL_EXHAUSTIVE,"exhaustive")):
            if bname == "reference" and trav[1] == "exhaustive" and which != "atrium": continue
            u8, f32, st = ds.render(cfg, W, H, traversal=trav[0], collect_stats=True)
            u8, f32, st2 = ds.render(cfg, W, H, traversal=trav[0])
            rays = st['rays_primary']+st['rays_shadow']
            print(f"  {bname} {cname} {trav[1]}: kernel {st2['kernel_ms']:.3f} ms -> {rays/st2['kernel_ms']/1e3:.1f} Mrays/s ; boxes/ray {st['boxes_tested']/rays:.1f} tris/ray {st['triangles_tested']/rays:.1f} algoGB/s {(32*st['boxes_tested']+36*st['triangles_tested'])/st2['trace_kernel_ms']/1e6:.1f}", flush=True)
            imgs[(bname,cname,trav[1])] = f32
            if trav[1]=="ordered": write_png(os.path.join(ROOT,"gpurun_out",f"{which}_{cname}_{bname}.png"), u8[::2, ::2] if which=="atrium" else u8[::8, ::8])
    ds.close()
keys = list(imgs)
for cname,_ in cfgs:
    ref = imgs[("reference",cname,"ordered")]
    for k in keys:
        if k[1]==cname:
            d = np.abs(imgs[k]-ref).max(-1)
            print(k, "vs reference/ordered: differing px", int((d>0).sum()), "max", float(d.max()), flush=True)
