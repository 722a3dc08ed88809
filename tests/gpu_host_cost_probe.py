"""Host-side cost of one frame of a rank's loop, piece by piece (us per call, GPU idle otherwise between pieces is fine: only the CPU time of
each call is taken).  usage: python tests/gpu_host_cost_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29534")
import torch
import torch.distributed as dist
from rayca_amd import Config, DeviceScene, flatten, scenes, abi
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
cfg, W, H = Config(max_depth=1), 1920, 1080
ds = DeviceScene(flatten(scenes.atrium_scene()), cfg, builder=abi.BUILDER_SAH)
tile = (0, 8, 8)
rows = ds.tile_rows(tile, H)
st, comm = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
send = torch.empty((rows, W, 4), dtype=torch.uint8, device=dev)
recv = [torch.empty((rows, W, 4), dtype=torch.uint8, device=dev)]
ev = torch.cuda.Event()
for _ in range(12):
    ds.render_device(cfg, W, H, send.data_ptr(), 0, tile=tile, stream=st.cuda_stream, want_stats=True)
issue = ds.prepare_device(cfg, W, H, send.data_ptr(), 0, tile=tile, stream=st.cuda_stream)
pg = dist.distributed_c10d._get_default_group()
opts = dist.GatherOptions()
opts.rootRank = 0
K = 300


def timed(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        fn()
    t = (time.perf_counter() - t0) / K * 1e6
    torch.cuda.synchronize()
    print(f"{name:48s} {t:7.1f} us", flush=True)


def gather_py():
    with torch.cuda.stream(comm):
        dist.gather(send, recv, dst=0)


def gather_pg():
    with torch.cuda.stream(comm):
        pg.gather([recv], [send], opts)


def wait_ev():
    with torch.cuda.stream(comm):
        comm.wait_event(ev)


timed("render_device (unprepared, ctypes marshalling)", lambda: ds.render_device(cfg, W, H, send.data_ptr(), 0, tile=tile, stream=st.cuda_stream))
timed("prepared issue()", issue)
timed("event.record(stream)", lambda: ev.record(st))
timed("with stream(comm): comm.wait_event", wait_ev)
timed("with stream(comm): dist.gather", gather_py)
timed("with stream(comm): ProcessGroup.gather", gather_pg)
dist.destroy_process_group()
