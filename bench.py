"""bench.py -- Mrays/s and ms/frame of the rayca hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload atrium|soup|cornell|box]

A "step" is one frame: every rank renders its rows of the frame with the HIP kernels (scene and BVH
already resident in HBM) and rank 0 receives the gathered RGBA8 frame (one RCCL gather per frame,
N > 1 only).  Default workload = the configuration BASELINE.json's metric is quoted on:
1920x1080, primary + 1 shadow ray per hit (Pathtracer max_depth=1, NEE, one point light), 1 spp, on
the ~272k-triangle procedural atrium -- a STAND-IN for Sponza, which is not available offline.

One JSON line on rank 0:
  value      whole-job Mrays/s = rays traced by all ranks in K steps / max-over-ranks wall time
  roofline   dominant kernel k_generation (generation 0: camera rays + traversal + shading + shadow
             rays): algorithmic bytes per launch (32 B per box test + 36 B per triangle test + 272 B
             per shaded hit + 4 B per pixel, counted by the instrumented variant of the same kernel)
             / mean launch duration from HIP events on the launch stream, against 8 TB/s HBM peak
  cpu_baseline  the CPU oracle (port of the reference algorithm, per-test vertex transforms, all host
             cores) on an evenly spaced subset of the same frame's rows; rank 0, N=1 only
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak (6.3 TB/s achievable)


def workload_config(name):
    from rayca_amd import Config, IntegratorStrategy
    if name == "atrium":
        return dict(label="sponza-STAND-IN procedural atrium 271,568 tris, 1920x1080, primary + 1 shadow ray (Pathtracer max_depth=1, NEE, 1 point light), 1 spp",
                    width=1920, height=1080, cfg=Config(max_depth=1))
    if name == "atrium4k":
        return dict(label="sponza-STAND-IN procedural atrium 271,568 tris, 3840x2160, 4-bounce path trace (Pathtracer max_depth=5, NEE + cosine), 1 spp",
                    width=3840, height=2160, cfg=Config(max_depth=5))
    if name == "soup":
        return dict(label="synthetic soup 1,048,576 random tris (seed 0x5EED0001), 4096x4096, primary rays only (Flat), 1 spp",
                    width=4096, height=4096, cfg=Config(integrator=IntegratorStrategy.Flat))
    if name == "cornell":
        return dict(label="cornell-style room 36 tris, 1920x1080, primary rays only (Flat), 1 spp",
                    width=1920, height=1080, cfg=Config(integrator=IntegratorStrategy.Flat))
    if name == "box":
        return dict(label="Khronos Box glTF 12 tris + default model, 256x256, Pathtracer max_depth=1", width=256, height=256,
                    cfg=Config(max_depth=1))
    raise SystemExit(f"unknown workload {name}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="atrium")
    ap.add_argument("--band-rows", type=int, default=8)
    ap.add_argument("--builder", default="sah", choices=["sah", "reference"],
                    help="sah: SAH tree with empty-seeded candidate boxes (default); reference: the reference's tree, quirks included")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="frames rendered concurrently (frame contexts + streams); 0 = 2 on one GPU, 4 (the HIP hardware queues) with more")
    ap.add_argument("--gather-batch", type=int, default=4,
                    help="N > 1: finished frames gathered per collective (1 = one gather per frame)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)   # one process per GPU; the modulo only matters for rehearsals
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # RCCL ("nccl") over xGMI is the real path; RAYCA_DIST_BACKEND=gloo only exists to rehearse the
    # N>1 control flow on a box with fewer GPUs than ranks (RCCL refuses two ranks on one device)
    backend = os.environ.get("RAYCA_DIST_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import __graft_entry__ as g
    if rank == 0:
        g.build()
    if world > 1:
        dist.barrier()
    from rayca_amd import DeviceScene, abi, flatten, scenes
    from rayca_amd.distributed import FrameGatherer, rows_of, tile_of

    wl = workload_config(args.workload)
    cfg, W, H = wl["cfg"], wl["width"], wl["height"]
    scene = scenes.WORKLOADS["atrium" if args.workload == "atrium4k" else args.workload]["scene"]()
    desc = flatten(scene)
    builder = abi.BUILDER_SAH if args.builder == "sah" else abi.BUILDER_REFERENCE
    ds = DeviceScene(desc, cfg, device=dev_index, builder=builder)
    info = ds.info()
    tile = tile_of(rank, world, args.band_rows)
    my_rows = int(rows_of(tile, H).numel())
    # F frames in flight: frame i renders with frame context i % F on its own stream into its own buffer, so the tail
    # of one frame (a few slow waves) overlaps the head of the next; with N > 1 the gather of a finished frame runs on
    # the comm stream meanwhile.  Every frame is still one complete pass: camera rays -> pixels (-> gather).
    # measured on one MI355X (tests/gpu_inflight_probe.py): a whole 1080p frame per GPU is best with 2 in flight (0.497 ms
    # against 0.656 with 1), a rank's half / quarter / eighth of it with 4 (0.262 / 0.146 / 0.094 ms); beyond 4 -- the
    # number of HIP hardware queues -- it gets worse again
    wl_generations = int(getattr(cfg, "max_depth", 1)) if int(getattr(cfg, "integrator", 5)) == 5 else 1
    F = args.frames_in_flight if args.frames_in_flight > 0 else (2 if world == 1 else 4)
    F = max(1, min(F, 8))
    outs = [torch.empty((my_rows, W, 4), dtype=torch.uint8, device=dev) for _ in range(F)]
    out = outs[0]
    streams = [torch.cuda.Stream(dev) for _ in range(F)]   # render kernels, one stream per frame in flight
    stream = streams[0]
    comm = torch.cuda.Stream(dev)                           # frame gather (RCCL)
    counter = [0]
    gather_dev = dev if backend == "nccl" else torch.device("cpu")
    # N > 1: finished frames are gathered B at a time with ONE collective (a rank's rows of a 1080p frame are 1 MB at
    # 8 ranks: per-call latency and the all-rank synchronisation of a collective cost more than its bytes).  Frames are
    # rendered straight into slot b of one of two batch buffers; the gather of a batch overlaps the next batch's rendering.
    B = max(1, min(args.gather_batch, 8)) if world > 1 else 1
    gatherer = FrameGatherer(H, W, args.band_rows, gather_dev, batch=B) if world > 1 else None
    sends = [torch.zeros((B, gatherer.max_rows, W, 4), dtype=torch.uint8, device=dev) for _ in range(2)] if world > 1 else []
    frames = [torch.empty((B, H, W, 4), dtype=torch.uint8, device=gather_dev) for _ in range(2)] if (world > 1 and rank == 0) else [None, None]
    ev_render = [[torch.cuda.Event() for _ in range(B)] for _ in range(2)]
    ev_gather = [torch.cuda.Event() for _ in range(2)]
    for e in ev_gather:
        e.record(comm)

    def flush(k, n):
        """gather the first n frames' worth of batch buffer k (always the whole buffer: one message shape)"""
        with torch.cuda.stream(comm):
            for e in ev_render[k][:n]:
                comm.wait_event(e)
            if backend == "nccl":
                got = gatherer.gather_batch(sends[k], frames[k])
            else:  # rehearsal: gloo gathers host tensors
                comm.synchronize()
                got = gatherer.gather_batch(sends[k].cpu(), frames[k])
            ev_gather[k].record(comm)
        return got

    def step(want_stats=False):
        i = counter[0]
        counter[0] += 1
        c = i % F
        st_c = streams[c]
        if world == 1:
            with torch.cuda.stream(st_c):
                st = ds.render_device(cfg, W, H, outs[c].data_ptr(), 0, tile=tile, stream=st_c.cuda_stream, want_stats=want_stats, context=c)
            return st, outs[c]
        b, k = i % B, (i // B) % 2
        buf = sends[k][b, :my_rows]
        with torch.cuda.stream(st_c):
            st_c.wait_event(ev_gather[k])   # the previous gather out of this batch buffer has finished
            st = ds.render_device(cfg, W, H, buf.data_ptr(), 0, tile=tile, stream=st_c.cuda_stream, want_stats=want_stats, context=c)
            ev_render[k][b].record(st_c)
        return st, (flush(k, B) if b == B - 1 else None)

    def drain():
        """gather what an incomplete batch holds and start the next step on a batch boundary"""
        if world > 1 and counter[0] % B:
            flush((counter[0] // B) % 2, counter[0] % B)
        counter[0] = 0

    # the scene times binary against 4-wide nodes on its first large frames (RaycaStats.node_format bit 8) and then
    # keeps the faster: let that finish before anything is timed
    node_format = 0
    for _ in range(20):
        st0 = ds.render_device(cfg, W, H, out.data_ptr(), 0, tile=tile, stream=stream.cuda_stream, want_stats=True)
        node_format = st0["node_format"]
        if not node_format & 256:
            break
    for i in range(1, F):   # first use of a frame context allocates its work buffers: not inside the timed region
        ds.render_device(cfg, W, H, outs[i].data_ptr(), 0, tile=tile, stream=streams[i].cuda_stream, context=i)
    torch.cuda.synchronize()
    # instrumented run with the node format just chosen: rays + algorithmic bytes of this rank's launch (not timed)
    counted = ds.render_device(cfg, W, H, out.data_ptr(), 0, tile=tile, stream=stream.cuda_stream, want_stats=True, collect_stats=True)
    rays_rank = counted["rays_primary"] + counted["rays_shadow"] + counted["rays_bounce"]
    algo_bytes = 32 * counted["boxes_tested"] + 36 * counted["triangles_tested"] + 272 * counted["hits_shaded"] + 4 * my_rows * W

    for _ in range(args.warmup):
        step()
    drain()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, frame = step()
    drain()   # every one of the K frames is gathered inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # kernel durations with HIP events on the launch stream (separate loop: the events force a sync per step)
    kms, tms = [], []
    for _ in range(max(3, min(args.steps, 20))):
        st = ds.render_device(cfg, W, H, out.data_ptr(), 0, tile=tile, stream=stream.cuda_stream, want_stats=True)
        kms.append(st["kernel_ms"])
        tms.append(st["trace_kernel_ms"] / max(st["trace_kernel_launches"], 1))
        launches_per_frame = max(int(st["trace_kernel_launches"]), 1)
    trace_ms = float(np.mean(tms))   # average duration of one traversal-kernel launch
    # algorithmic bytes are counted over the frame: per launch they are the frame's bytes / its traversal launches
    # (1 for the benchmark frame; a 4-bounce frame has 5 generations x (trace + shadow))
    algo_bytes = algo_bytes / launches_per_frame

    t = torch.tensor([elapsed, float(rays_rank), float(algo_bytes), trace_ms], dtype=torch.float64,
                     device=dev if backend == "nccl" else "cpu")
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed_max, rays_total = float(tmax[0]), float(tsum[1])
    else:
        elapsed_max, rays_total = elapsed, float(rays_rank)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    mrays = rays_total * args.steps / elapsed_max / 1e6
    achieved = algo_bytes / (trace_ms * 1e-3) / 1e9
    result = {
        "metric": "Mrays/sec (primary+shadow), 1920x1080 Sponza 1spp" if args.workload == "atrium" else f"Mrays/sec ({args.workload})",
        "value": round(mrays, 3),
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed_max / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": wl["label"], "width": W, "height": H, "triangles": info["triangle_count"],
                   "bvh_nodes": info["node_count"], "bvh_build_ms": round(info["build_ms"], 1), "bvh_built_on": "gpu (bvh_build.hip; same tree as the host builder)",
                   "node_format": {"generation0": ("4-wide" if node_format & 1 else "binary") + (" fp16" if node_format & 4 else " f32"),
                                   "bounces": ("4-wide" if node_format & 2 else "binary") + (" fp16" if node_format & 8 else " f32"),
                                   "chosen_by": "timing the four formats on this scene (same pixels with each)"},
                   "bvh_builder": ("SAH 63 planes x 3 axes as rayca-soft bvh/blas.rs, candidate boxes seeded empty; ties by the reference's primitive order"
                                   if args.builder == "sah" else "reference SAH (rayca-soft bvh/blas.rs:64-123,261-316) incl. origin-seeded candidate boxes"),
                   "rays_per_frame": int(rays_total), "frames_in_flight": F, "tiling": f"rows in bands of {args.band_rows} dealt over {world} rank(s)",
                   "frame_gather": (f"torch.distributed.gather ({'RCCL' if backend == 'nccl' else backend}) to rank 0, {B} finished frame(s) per collective, overlapped with rendering"
                                    if world > 1 else "none")},
        "roofline": {"bound": "hbm", "kernel": ("k_wf_trace / k_wf_shadow (all generations; wavefront engine from three generations up)" if wl_generations >= 3 else "k_generation (generation 0)"), "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                     "algorithmic_bytes_per_launch": int(algo_bytes), "launch_ms": round(trace_ms, 4), "launches_per_frame": launches_per_frame,
                     "boxes_tested": int(counted["boxes_tested"]), "triangles_tested": int(counted["triangles_tested"]),
                     "hits_shaded": int(counted["hits_shaded"]), "frame_kernel_ms": round(float(np.mean(kms)), 4)},
    }
    # HBM traffic cannot be read from inside this process: it comes from the committed rocprofv3 --pmc
    # passes over this same workload (profiles/pmc_traffic_<workload>.json), per launch like `achieved`
    pmc = os.path.join(ROOT, "profiles", f"pmc_traffic_{args.workload}.json")
    if world == 1 and args.builder == "sah" and os.path.exists(pmc):
        traffic = json.load(open(pmc))["traffic_bytes_per_launch"]
        result["roofline"]["traffic"] = traffic
        result["roofline"]["traffic_source"] = os.path.relpath(pmc, ROOT)
        # `frac` prices every node and triangle a ray touches against HBM; the trees of these scenes are L2/MALL
        # resident, so what actually crosses the HBM interface is this much smaller fraction of peak
        result["roofline"]["traffic_frac_of_peak"] = round(traffic / (trace_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        result["roofline"]["note"] = ("working set is cache resident: frac > 1 means algorithmic bytes are served by L2/MALL; the kernel is bound by "
                                      "dependent-fetch latency (PMC: profiles/*pmc_summary.json, DESIGN.md section 8)")
    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(desc, cfg, W, H, args.cpu_seconds)
    print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(desc, cfg, W, H, budget_s):
    """The oracle (test infrastructure, a port of the reference algorithm) timed on this box's cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    # the GPU box gives a one-GPU job a share of 16 host cores (the machine has more); use that share
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = int(os.environ.get("RAYCA_CPU_THREADS", min(avail, 16)))
    orc = ol.OracleScene(desc, cfg, build=ol.BUILD_BINNED, xform=ol.XFORM_PER_TEST, threads=cores)
    # probe: every `parts`-th row, starting sparse
    parts = max(H // 8, 1)
    _, _, st = orc.render(cfg, W, H, tile=(0, parts, 1), want_rgba8=False, want_f32=False)
    rows_probe = st["rows_rendered"]
    per_row = st["seconds"] / max(rows_probe, 1)
    want_rows = int(min(H, max(rows_probe, budget_s / max(per_row, 1e-9))))
    parts = max(H // want_rows, 1)
    _, _, st = orc.render(cfg, W, H, tile=(0, parts, 1), want_rgba8=False, want_f32=False)
    rays = st["rays_primary"] + st["rays_shadow"] + st["rays_bounce"]
    return {"value": round(rays / st["seconds"] / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"every {parts}-th row of the same {W}x{H} frame ({st['rows_rendered']} rows, {rays} rays, "
                      f"{st['seconds']:.2f} s wall); per-test vertex transforms as in the reference; BVH build excluded "
                      "like the reference's own timer (scene.rs:101,152)"}


if __name__ == "__main__":
    main()
