"""What one rank of an N-GPU run costs per frame on THIS GPU: render part 0 of `parts` (bands of 8 rows) with F frames in
flight, optionally followed by a per-frame RCCL gather on a 1-rank group (the collective's launch + Python cost without the
wire and without the other ranks' skew), with the loop bench.py's ranks run: ONE native call per frame (wait for the previous
gather out of the buffer, render, record the frame's event), the collective on the comm stream.
usage: python tests/gpu_rank_share_probe.py [atrium|atrium4k] [parts ...]     RAYCA_PROBE_F=3,4  RAYCA_PROBE_CALLS=1|3"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np
import torch
import torch.distributed as dist
import bench
from rayca_amd import DeviceScene, flatten, scenes, abi
from rayca_amd.streams import frame_streams
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
args = sys.argv[1:]
wl = args.pop(0) if args and not args[0].isdigit() else "atrium"
w = bench.workload_config(wl)
cfg, W, H = w["cfg"], w["width"], w["height"]
fs, sp = frame_streams(dev, 8)
early_streams = fs + sp
ds = DeviceScene(flatten(scenes.atrium_scene()), cfg, builder=abi.BUILDER_SAH)
ds.finish()
parts_list = [int(a) for a in args] or [1, 2, 4, 8]
single_call = os.environ.get("RAYCA_PROBE_CALLS", "1") == "1"
K = 200 if wl == "atrium" else 24
whole_ms = None
for parts in parts_list:
    tile = (0, parts, 8)
    rows = ds.tile_rows(tile, H)
    for F in [int(x) for x in os.environ.get("RAYCA_PROBE_F", "4").split(",")]:
        streams = early_streams[8 - F:8]
        comm = early_streams[8]
        # RAYCA_PROBE_BUFS send buffers per frame context, used in turn: with one, a context's next frame waits for the gather of
        # its previous one (render -> gather -> render is one dependent chain per context); with two it only waits for the render
        NB = F * int(os.environ.get("RAYCA_PROBE_BUFS", "1"))
        sends = [torch.empty((rows, W, 4), dtype=torch.uint8, device=dev) for _ in range(NB)]
        recv = [torch.empty((rows, W, 4), dtype=torch.uint8, device=dev)]
        ev = [torch.cuda.Event() for _ in range(NB)]
        evg = [torch.cuda.Event() for _ in range(NB)]
        for e in ev + evg:
            e.record(comm)
        for i in range(F):   # contexts warm, node format decided
            for _ in range(10 if wl == "atrium" else 3):
                ds.render_device(cfg, W, H, sends[i].data_ptr(), 0, tile=tile, stream=streams[i].cuda_stream, context=i, want_stats=True)
        torch.cuda.synchronize()
        if single_call:
            issue = [ds.prepare_device(cfg, W, H, sends[i].data_ptr(), 0, tile=tile, stream=streams[i % F].cuda_stream, context=i % F,
                                       wait_event=evg[i].cuda_event, record_event=ev[i].cuda_event) for i in range(NB)]
        else:
            issue = [ds.prepare_device(cfg, W, H, sends[i].data_ptr(), 0, tile=tile, stream=streams[i % F].cuda_stream, context=i % F) for i in range(NB)]
        comm_thread = os.environ.get("RAYCA_PROBE_COMM_THREAD", "0") == "1"
        on_frame_stream = os.environ.get("RAYCA_PROBE_COMM", "") == "frame"
        if on_frame_stream:   # (stream order is all the ordering there is: no events)
            issue = [ds.prepare_device(cfg, W, H, sends[i].data_ptr(), 0, tile=tile, stream=streams[i % F].cuda_stream, context=i % F) for i in range(NB)]
            single_call_here = True
        if comm_thread:   # the collectives issued by a thread of their own: the frame loop only hands it frame indices
            import queue, threading
            jobs = queue.SimpleQueue()

            def comm_loop():
                torch.cuda.set_device(0)
                while True:
                    i = jobs.get()
                    if i is None:
                        return
                    with torch.cuda.stream(comm):
                        comm.wait_event(ev[i])
                        dist.gather(sends[i], recv, dst=0)
                        evg[i].record(comm)
            th = threading.Thread(target=comm_loop, daemon=True)
            th.start()
            # (a frame's wait_event must see the gather's record: the thread records evg[i] some time after the frame loop has
            # moved on -- with F buffers in turn the loop comes back to buffer i F frames later; guarded by a host-side count)
            done = [threading.Semaphore(0) for _ in range(F)]
        for gather in (False, True):
            host = 0.0
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if comm_thread and gather:
                pending = [0] * F

                def comm_loop2():
                    torch.cuda.set_device(0)
                    while True:
                        i = jobs.get()
                        if i is None:
                            return
                        with torch.cuda.stream(comm):
                            comm.wait_event(ev[i])
                            dist.gather(sends[i], recv, dst=0)
                            evg[i].record(comm)
                        done[i].release()
                jobs.put(None)
                th.join()
                th = threading.Thread(target=comm_loop2, daemon=True)
                th.start()
                for k in range(K):
                    i = k % F
                    h0 = time.perf_counter()
                    if pending[i]:
                        done[i].acquire()      # the gather out of this buffer has been ISSUED (its event recorded): the frame may wait on it
                        pending[i] = 0
                    issue[i]()
                    jobs.put(i)
                    pending[i] = 1
                    host += time.perf_counter() - h0
                for i in range(F):
                    if pending[i]:
                        done[i].acquire()
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / K * 1e3
                scale = f"x{whole_ms / ms:.2f} of one GPU's" if whole_ms else ""
                print(f"{wl} parts {parts} ({rows} rows) F={F} comm thread     gather=per frame: {ms:.4f} ms/frame, host (frame loop) {host / K * 1e6:.1f} us/frame  -> {parts}-GPU frame rate {scale}", flush=True)
                continue
            for k in range(K):
                i = k % NB
                h0 = time.perf_counter()
                if single_call or on_frame_stream:
                    issue[i]()
                else:
                    streams[i % F].wait_event(evg[i])
                    issue[i]()
                    ev[i].record(streams[i % F])
                if gather and on_frame_stream:   # RAYCA_PROBE_COMM=frame: the collective behind the frame on the frame's own stream
                    with torch.cuda.stream(streams[i % F]):
                        dist.gather(sends[i], recv, dst=0)
                elif gather:
                    with torch.cuda.stream(comm):
                        comm.wait_event(ev[i])
                        dist.gather(sends[i], recv, dst=0)
                        evg[i].record(comm)
                host += time.perf_counter() - h0
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / K * 1e3
            if parts == 1 and not gather:
                whole_ms = ms   # one GPU's own frame time, measured here: what the shares are compared with
            scale = f"x{whole_ms / ms:.2f} of one GPU's" if whole_ms else "(run with parts 1 first for the ratio)"
            print(f"{wl} parts {parts} ({rows} rows) F={F} bufs={NB} calls/frame={'1' if single_call else '3'}{' gather on the frame stream' if on_frame_stream else ''} gather={'per frame' if gather else 'none':9s}: {ms:.4f} ms/frame, host {host / K * 1e6:.1f} us/frame  -> {parts}-GPU frame rate {scale}", flush=True)
if os.environ.get("RAYCA_PROBE_COMM_THREAD", "0") == "1":
    jobs.put(None)
dist.destroy_process_group()
