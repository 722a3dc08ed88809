"""bench.py -- Mrays/s and ms/frame of the rayca hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload atrium|soup|cornell|box|atrium4k] [--no-others]

A "step" is one frame: every rank renders its rows of the frame with the HIP kernels (scene and BVH already resident
in HBM) and rank 0 receives the gathered RGBA8 frame (N > 1: ONE RCCL gather per frame, at frame end).  Default
workload = the configuration BASELINE.json's metric is quoted on: 1920x1080, primary + 1 shadow ray per hit (Pathtracer
max_depth=1, NEE, one point light), 1 spp, on the ~272k-triangle procedural atrium -- a STAND-IN for Sponza, which is
not available offline.  The default run then times BASELINE's other GPU configurations the same way, shorter, and
reports them under `config.other_workloads` (each with its own roofline): the 1M-triangle soup at 4096x4096 (Flat), the
Cornell room at 1080p (Flat), the atrium at 3840x2160 with four bounces.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (rayca_amd/launcher.py: plain
child processes, started before anything touches a GPU); under `python -m torch.distributed.run` the ranks come from
the launcher's environment as usual.

One JSON line on rank 0:
  value         whole-job Mrays/s = rays traced by all ranks in K steps / max-over-ranks wall time, with ONE gather per
                frame for N > 1 (`config.frame_gather`); the same K steps with four frames per collective are timed as
                well and reported next to it (`config.gather_batched`) -- never as `value`
  roofline      the kernel the frame spends most of its time in (RaycaStats.class_ms: HIP events around every launch,
                on the launch stream, by kernel class): its average launch duration, priced against every hardware limit
                it could hit -- VALU issue, L1 (vector cache) accesses, L2 fill, HBM traffic, compulsory HBM bytes; `bound`
                names the one it is closest to and `frac` is that fraction.  The per-launch counter values come from the
                committed rocprofv3 --pmc passes of this same command (profiles/pmc_counters_<workload>.json: used only
                when they describe the same kernel on the same node format), the two peaks that are not in the hardware
                guide -- VALU issue rate and L1 accesses per clock -- from the microbenchmark tests/microbench/peaks.hip
                (profiles/peaks_r03.json).  SURVEY 8(d)'s algorithmic bytes (32 B per box test + 36 B per triangle test
                + 272 B per shaded hit + 4 B per pixel, counted by the instrumented instantiation of the same kernels)
                are kept as `algorithmic_access_rate`: they are served by L1/L2, not by HBM.
  latency       N = 1: one frame at a time -- kernels only (HIP events), through rayca_hip_render_device + a synchronise,
                and through rayca_hip_render, the drop-in draw(): kernels + the copy into the caller's host image
                (what scene.rs:101-152 times), into ordinary and into page-locked host memory
  cpu_baseline  the CPU oracle (port of the reference algorithm, per-test vertex transforms, all host cores of this
                job's share) on an evenly spaced subset of the same frame's rows; rank 0, N = 1, main workload only
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0          # HBM3E 8 TB/s peak (6.3 TB/s achievable)
L2_PEAK_GBS = 34500.0          # L2 aggregate, 8 XCDs
CLOCK_HZ = 2.4e9               # max engine clock
CUS, SIMDS = 256, 1024
# the two ceilings the guide does not give, as measured by tests/microbench/peaks.hip (profiles/peaks_r03.json replaces
# these when present): SIMD cycles per wave64 VALU instruction on the slab test's own instruction mix at four waves per
# SIMD, and vector-L1 accesses (TCP_TOTAL_CACHE_ACCESSES: one 128-B line of one instruction, or 64 B of a coherent one)
# per clock and CU
VALU_MIX_CYCLES_DEFAULT = 3.6
L1_ACCESSES_PER_CLK_CU_DEFAULT = 0.99

KERNEL_NAMES = ("k_generation", "k_flat_refill", "k_wf_trace", "k_queue_refill", "k_wf_shade", "k_wf_shadow", "k_shadow_refill", "other")

# the other BASELINE GPU configurations the default run times after the main one: (workload, steps, warmup)
OTHER_WORKLOADS = (("soup", 24, 4), ("cornell", 200, 8), ("atrium4k", 16, 4))


def workload_config(name):
    from rayca_amd import Config, IntegratorStrategy
    if name == "atrium":
        return dict(label="sponza-STAND-IN procedural atrium 271,568 tris, 1920x1080, primary + 1 shadow ray (Pathtracer max_depth=1, NEE, 1 point light), 1 spp",
                    baseline_config="configs[2]", width=1920, height=1080, cfg=Config(max_depth=1))
    if name == "atrium4k":
        return dict(label="sponza-STAND-IN procedural atrium 271,568 tris, 3840x2160, 4-bounce path trace (Pathtracer max_depth=5, NEE + cosine), 1 spp",
                    baseline_config="configs[4]", width=3840, height=2160, cfg=Config(max_depth=5))
    if name == "soup":
        return dict(label="synthetic soup 1,048,576 random tris (seed 0x5EED0001), 4096x4096, primary rays only (Flat), 1 spp",
                    baseline_config="configs[3]", width=4096, height=4096, cfg=Config(integrator=IntegratorStrategy.Flat))
    if name == "cornell":
        return dict(label="cornell-style room 36 tris, 1920x1080, primary rays only (Flat), 1 spp",
                    baseline_config="configs[1]", width=1920, height=1080, cfg=Config(integrator=IntegratorStrategy.Flat))
    if name == "box":
        return dict(label="Khronos Box glTF 12 tris + default model, 256x256, Pathtracer max_depth=1", baseline_config="configs[0]", width=256, height=256,
                    cfg=Config(max_depth=1))
    raise SystemExit(f"unknown workload {name}")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)   # (0.4 ms each: long enough that filling and draining four frames in flight is noise)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default="atrium")
    ap.add_argument("--no-others", action="store_true", help="only the main workload (the default run adds soup, cornell and atrium4k under config.other_workloads)")
    ap.add_argument("--band-rows", type=int, default=8)
    ap.add_argument("--builder", default="sah", choices=["sah", "reference"],
                    help="sah: SAH tree with empty-seeded candidate boxes (default); reference: the reference's tree, quirks included")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="frames rendered concurrently (frame contexts + streams); 0 = 4 on one GPU (the number of HIP hardware queues), 3 per rank with N > 1 (the fourth queue is the gather's)")
    ap.add_argument("--gather-batch", type=int, default=4,
                    help="N > 1: frames per collective of the SECOND timed run (the first, `value`, always gathers every frame on its own)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args(argv)


def measured_peaks():
    """VALU issue and L1 ceilings from the committed microbenchmark summary, or the defaults above."""
    path = os.path.join(ROOT, "profiles", "peaks_r03.json")
    valu, l1, src = VALU_MIX_CYCLES_DEFAULT, L1_ACCESSES_PER_CLK_CU_DEFAULT, None
    if os.path.exists(path):
        try:
            p = json.load(open(path))
            valu = float(p["valu"]["traversal_mix_cycles_per_instr"])
            l1 = float(p["l1"]["accesses_per_clk_per_cu"])
            src = os.path.relpath(path, ROOT)
        except (KeyError, ValueError, TypeError):
            pass
    return valu, l1, src


def pmc_for(workload, builder, kernel, node_format_bits, record_bytes=64):
    """The committed per-launch counters of `kernel` on this workload -- only if they were collected on the same node
    format (generation kernels: the scene times the formats itself and may settle on another one than the profiled run).
    Returns (counters dict, launch_ms_under_pmc | None, source path) or (None, None, why)."""
    path = os.path.join(ROOT, "profiles", f"pmc_counters_{workload}.json")
    if builder != "sah":
        return None, None, "counters were collected on the SAH tree only"
    if not os.path.exists(path):
        return None, None, f"no PMC counter file for {workload}"
    pmc = json.load(open(path))
    if "kernels" in pmc:
        k = pmc["kernels"].get(kernel)
        if not k:
            return None, None, f"{os.path.basename(path)} has no counters for {kernel}"
    else:   # (round-2 layout: one kernel per file)
        if pmc.get("kernel") != kernel:
            return None, None, f"{os.path.basename(path)} describes {pmc.get('kernel')}, the frame runs on {kernel}"
        k = pmc
    want = k.get("node_format_bits")
    if want is not None and kernel in ("k_generation", "k_flat_refill") and int(want) != int(node_format_bits):
        return None, None, f"counters were collected on node format bits {want}, this run settled on {node_format_bits}"
    if int(k.get("binary_f32_record_bytes", 64)) != int(record_bytes):   # (the binary f32 node record the library was built with)
        return None, None, f"counters were collected on {k.get('binary_f32_record_bytes', 64)}-B binary f32 node records, this library reads {record_bytes}-B ones"
    return k["counters_per_launch"], k.get("launch_ms_under_pmc"), os.path.relpath(path, ROOT)


def roofline_limits(c, launch_ms, compulsory_bytes):
    """Fractions of the hardware limits a kernel runs at: per-launch PMC counters `c` over the launch duration measured live."""
    valu_cycles, l1_rate, peaks_src = measured_peaks()
    t = launch_ms * 1e-3
    gbs = lambda b: b / t / 1e9   # noqa: E731
    hbm_bytes = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0   # KiB; gfx950: FETCH_SIZE tallies 128-B requests at 64 B
    l1_peak = CUS * CLOCK_HZ * l1_rate / 1e9     # G accesses/s
    limits = {
        "valu_issue": {"achieved": round(c["SQ_INSTS_VALU"] / t / 1e9, 2), "peak": round(SIMDS * CLOCK_HZ / valu_cycles / 1e9, 1), "unit": "G wave-instr/s",
                       "how": f"SQ_INSTS_VALU / launch time; peak = 1024 SIMDs x 2.4 GHz / {valu_cycles:.2f} cycles, the rate measured on the slab test's own instruction mix "
                              "(v_fma_f32 issues in ~2.4 cycles with two or more waves per SIMD, min / max / cmp / cndmask / fma_mix in 4.1: tests/microbench/peaks.hip)"},
        "l1_accesses": {"achieved": round(c["TCP_TOTAL_CACHE_ACCESSES_sum"] / t / 1e9, 2), "peak": round(l1_peak, 1), "unit": "G accesses/s",
                        "how": f"TCP_TOTAL_CACHE_ACCESSES / launch time; peak = {l1_rate:.2f} per clock and CU x 256 CUs x 2.4 GHz, measured: one access = one 128-B line of one "
                               "instruction (divergent lanes) or 64 B of a coherent one, and the vector L1 retires one per clock whatever the pattern (tests/microbench/peaks.hip)"},
        "l2_fill": {"achieved": round(gbs(c["TCP_TCC_READ_REQ_sum"] * 64.0), 1), "peak": L2_PEAK_GBS, "unit": "GB/s",
                    "how": "TCP_TCC_READ_REQ x 64 B / launch time"},
        "hbm_traffic": {"achieved": round(gbs(hbm_bytes), 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "how": "(2 x FETCH_SIZE + WRITE_SIZE) KiB from separate --pmc passes / launch time"},
        "hbm_compulsory": {"achieved": round(gbs(compulsory_bytes), 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "how": "scene (nodes + triangles + shading records) read once + path records + pixels written once / launch time"},
    }
    for v in limits.values():
        v["frac"] = round(v["achieved"] / v["peak"], 4)
    lanes = c.get("SQ_THREAD_CYCLES_VALU", 0.0) / max(64.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0), 1.0)
    return {"limits": limits, "valu_lane_activity": round(lanes, 3), "hbm_traffic_bytes": int(hbm_bytes), "peaks_source": peaks_src}


class Env:
    pass


def run_workload(env, args, name, steps, warmup, main):
    """Build the scene of one workload, time `steps` frames the way the driver's contract asks (warm-up, barrier +
    synchronise on both sides, max over ranks), measure the kernels one frame at a time, and return the pieces of the line."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from rayca_amd import DeviceScene, abi, flatten, scenes
    from rayca_amd.distributed import FrameGatherer, rows_of, tile_of

    rank, world, dev, dev_index, backend = env.rank, env.world, env.dev, env.dev_index, env.backend
    wl = workload_config(name)
    cfg, W, H = wl["cfg"], wl["width"], wl["height"]
    scene = scenes.WORKLOADS["atrium" if name == "atrium4k" else name]["scene"]()
    desc = flatten(scene)
    builder = abi.BUILDER_SAH if args.builder == "sah" else abi.BUILDER_REFERENCE
    ds = DeviceScene(desc, cfg, device=dev_index, builder=builder)
    info = ds.info()
    tile = tile_of(rank, world, args.band_rows)
    my_rows = int(rows_of(tile, H).numel())
    # F frames in flight: frame i renders with frame context i % F on its own stream into its own buffer, so the tail
    # of one frame (a few slow waves) overlaps the head of the next; with N > 1 the gather of a finished frame follows it on
    # the same stream while the other streams render.  Every frame is still one complete pass: camera rays -> pixels (-> gather).
    # measured on one MI355X with every frame stream on a hardware queue of its own (tests/gpu_inflight_probe.py, ms per frame
    # for 1 / 2 / 3 / 4 / 8 frames in flight): whole 1080p frame 0.538 / 0.420 / 0.412 / 0.404 / 0.406, a rank's half
    # 0.334 / 0.256 / 0.228 / 0.220 / 0.220, quarter 0.240 / 0.161 / 0.131 / 0.121 / 0.121, eighth 0.206 / 0.120 / 0.087 /
    # 0.071 / 0.071 -- four, the number of HIP hardware queues (eight queues, GPU_MAX_HW_QUEUES=8, change nothing)
    is_path = int(getattr(cfg, "integrator", 5)) == 5
    wl_generations = int(getattr(cfg, "max_depth", 1)) if is_path else 1
    # N > 1: four frame streams too, and the per-frame gather goes BEHIND ITS FRAME ON THE FRAME'S OWN STREAM (StreamGatherLoop):
    # stream order is all the ordering there is.  A comm stream of its own shares a hardware queue with one of four frame
    # streams (or leaves only three for the frames) and costs two event hops per frame: a rank's quarter / eighth of the 1080p
    # frame with the gather takes 0.099 / 0.062 ms per frame this way, against 0.124 / 0.088 with three frame streams + a
    # comm stream and 0.135 / 0.095 with four + one (tests/gpu_rank_share_probe.py, profiles/r03_rank_share.log; without a
    # gather: 0.097 / 0.060), and the rank's host spends 29 us per frame instead of 42-56.
    F = args.frames_in_flight if args.frames_in_flight > 0 else 4
    F = max(1, min(F, 8))
    outs = [torch.empty((my_rows, W, 4), dtype=torch.uint8, device=dev) for _ in range(F)]
    out = outs[0]
    streams = env.frame_streams[8 - F:]                     # render kernels, one stream per frame in flight
    stream = streams[0]
    comm = env.comm                                         # (made with them: the frame gather, RCCL)
    gather_dev = dev if backend == "nccl" else torch.device("cpu")

    def event():
        """a torch event that has a handle (torch makes it at the first record): its cuda_event can go into a prepared call"""
        e = torch.cuda.Event()
        e.record(comm)
        return e

    class GatherLoop:
        """K frames, B finished frames per collective.  B = 1 is the north-star shape: one gather per frame, at frame
        end.  Frames are rendered straight into slot b of one of two send buffers; the gather of one buffer runs on the
        comm stream and overlaps the rendering into the other.  Per frame the rank's host issues ONE native call -- wait for
        the previous gather out of this buffer, render, record the frame's event (RaycaRenderOptions.wait_event /
        record_event) -- and, when a batch is complete, the collective."""

        def __init__(self, B):
            self.B = B
            self.counter = 0
            self.issue = {}   # (send buffer, slot, context) -> prepared render call
            self.gatherer = FrameGatherer(H, W, args.band_rows, gather_dev, batch=B)
            self.sends = [torch.zeros((B, self.gatherer.max_rows, W, 4), dtype=torch.uint8, device=dev) for _ in range(2)]
            self.frames = [torch.empty((B, H, W, 4), dtype=torch.uint8, device=gather_dev) if rank == 0 else None for _ in range(2)]
            self.ev_render = [[event() for _ in range(B)] for _ in range(2)]
            self.ev_gather = [event() for _ in range(2)]
            torch.cuda.synchronize()

        def flush(self, k, n):
            with torch.cuda.stream(comm):
                for e in self.ev_render[k][:n]:
                    comm.wait_event(e)
                if backend == "nccl":
                    got = self.gatherer.gather_batch(self.sends[k], self.frames[k])
                else:  # rehearsal: gloo gathers host tensors
                    comm.synchronize()
                    got = self.gatherer.gather_batch(self.sends[k].cpu(), self.frames[k])
                self.ev_gather[k].record(comm)
            return got

        def step(self):
            i = self.counter
            self.counter += 1
            c = i % F
            b, k = i % self.B, (i // self.B) % 2
            issue = self.issue.get((k, b, c))
            if issue is None:
                buf = self.sends[k][b, :my_rows]
                issue = self.issue[(k, b, c)] = ds.prepare_device(cfg, W, H, buf.data_ptr(), 0, tile=tile, stream=streams[c].cuda_stream, context=c,
                                                                  wait_event=self.ev_gather[k].cuda_event, record_event=self.ev_render[k][b].cuda_event)
            issue()   # (the previous gather out of this send buffer has finished -> render -> the frame's event, one call)
            return self.flush(k, self.B) if b == self.B - 1 else None

        def drain(self):
            """gather what an incomplete batch holds and start the next step on a batch boundary"""
            if self.counter % self.B:
                self.flush((self.counter // self.B) % 2, self.counter % self.B)
            self.counter = 0

    class StreamGatherLoop:
        """K frames, one gather per frame (the north-star shape), issued behind the frame on the frame's stream: context c
        renders into its own send buffer, the collective and (rank 0) the de-interleave follow in stream order, the next frame
        of that context queues behind them.  No comm stream, no events; the F streams run F such chains side by side.  The
        collectives are issued in the same round-robin order on every rank."""
        B = 1

        def __init__(self):
            self.counter = 0
            # (one gatherer per context: each has a receive buffer of its own on rank 0 -- the de-interleave of one context's
            # frame runs on that context's stream while the next context's collective may already be landing)
            self.gatherers = [FrameGatherer(H, W, args.band_rows, gather_dev) for _ in range(F)]
            self.gatherer = self.gatherers[0]
            self.sends = [torch.zeros((self.gatherer.max_rows, W, 4), dtype=torch.uint8, device=dev) for _ in range(F)]
            self.frames = [torch.empty((H, W, 4), dtype=torch.uint8, device=gather_dev) if rank == 0 else None for _ in range(F)]
            self.issue = [ds.prepare_device(cfg, W, H, self.sends[c][:my_rows].data_ptr(), 0, tile=tile, stream=streams[c].cuda_stream, context=c) for c in range(F)]
            torch.cuda.synchronize()

        def step(self):
            c = self.counter % F
            self.counter += 1
            self.last = c
            self.issue[c]()
            with torch.cuda.stream(streams[c]):
                if backend == "nccl":
                    return self.gatherers[c](self.sends[c], out=self.frames[c])
                streams[c].synchronize()   # rehearsal: gloo gathers host tensors
                return self.gatherers[c](self.sends[c].cpu(), out=self.frames[c])

        def drain(self):
            self.counter = 0

    class LocalLoop:
        """N = 1: no exchange, the frame stays in device memory."""
        B = 1

        def __init__(self):
            self.counter = 0
            self.issue = [ds.prepare_device(cfg, W, H, outs[c].data_ptr(), 0, tile=tile, stream=streams[c].cuda_stream, context=c) for c in range(F)]

        def step(self):
            c = self.counter % F
            self.counter += 1
            self.issue[c]()
            return outs[c]

        def drain(self):
            self.counter = 0

    def timed(loop):
        """W untimed warm-up steps, then exactly K steps between barrier + synchronize on both sides."""
        for _ in range(warmup):
            loop.step()
        loop.drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loop.step()
        loop.drain()   # every one of the K frames is gathered inside the timed region
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    # the scene times binary against 4-wide nodes on its first large frames (RaycaStats.node_format bit 8) and then
    # keeps the faster: let that finish before anything is timed
    ds.finish()
    node_format = 0
    settle_deadline = time.perf_counter() + 10.0
    while time.perf_counter() < settle_deadline:
        st0 = ds.render_device(cfg, W, H, out.data_ptr(), 0, tile=tile, stream=stream.cuda_stream, want_stats=True)
        node_format = st0["node_format"]
        if not node_format & (256 | 512 | 2048):   # 256: still timing node formats, 512: the camera-ray kernels, 2048: formats still being made
            break
    for i in range(1, F):   # first use of a frame context allocates its work buffers: not inside the timed region
        ds.render_device(cfg, W, H, outs[i].data_ptr(), 0, tile=tile, stream=streams[i].cuda_stream, context=i)
    torch.cuda.synchronize()
    # instrumented run with the node format just chosen: rays + algorithmic bytes of this rank's launch (not timed)
    counted = ds.render_device(cfg, W, H, out.data_ptr(), 0, tile=tile, stream=stream.cuda_stream, want_stats=True, collect_stats=True)
    rays_rank = counted["rays_primary"] + counted["rays_shadow"] + counted["rays_bounce"]
    algo_bytes = 32 * counted["boxes_tested"] + 36 * counted["triangles_tested"] + 272 * counted["hits_shaded"] + 4 * my_rows * W

    frame_check = None
    if world == 1:
        elapsed = timed(LocalLoop())
        elapsed_batched, B2 = None, 1
    else:
        sg = StreamGatherLoop()
        elapsed = timed(sg)                                  # `value`: one gather per frame
        if rank == 0:   # the frame rank 0 holds after the last gather == the frame one GPU renders on its own, bit for bit
            whole = torch.empty((H, W, 4), dtype=torch.uint8, device=dev)
            ds.render_device(cfg, W, H, whole.data_ptr(), 0, stream=stream.cuda_stream)
            torch.cuda.synchronize()
            frame_check = bool(torch.equal(sg.frames[sg.last].to(dev), whole))
        B2 = max(1, min(args.gather_batch, 8))
        elapsed_batched = timed(GatherLoop(B2)) if (B2 > 1 and main) else None

    # kernel durations with HIP events on the launch stream, one frame at a time (separate loop: the events force a sync
    # per step), by kernel class: RaycaStats.class_ms / class_launches
    kms, tms = [], []
    class_ms = np.zeros(8)
    class_n = np.zeros(8)
    launches_per_frame = 1
    for _ in range(4):   # (the scene sizes its grids by the frame contexts of the last four calls: from here on there is one)
        ds.render_device(cfg, W, H, out.data_ptr(), 0, tile=tile, stream=stream.cuda_stream, want_stats=True)
    n_probe = max(3, min(steps, 20))
    for _ in range(n_probe):
        st = ds.render_device(cfg, W, H, out.data_ptr(), 0, tile=tile, stream=stream.cuda_stream, want_stats=True)
        kms.append(st["kernel_ms"])
        tms.append(st["trace_kernel_ms"])
        class_ms += np.array(st["class_ms"])
        class_n += np.array(st["class_launches"])
        launches_per_frame = max(int(st["trace_kernel_launches"]), 1)
    class_ms /= n_probe
    class_n /= n_probe
    frame_kernel_ms = float(np.mean(kms))
    trace_frame_ms = float(np.mean(tms))     # all traversal launches of a frame
    traversal = [k for k in range(7) if k != 4 and class_n[k] > 0]   # (4 = k_wf_shade, 7 = other: not traversal kernels)
    dominant = max(traversal, key=lambda k: class_ms[k]) if traversal else 7
    dom_name = KERNEL_NAMES[dominant]
    dom_launch_ms = float(class_ms[dominant] / max(class_n[dominant], 1.0))
    kernels = {KERNEL_NAMES[k]: {"launches_per_frame": round(float(class_n[k]), 2), "avg_launch_ms": round(float(class_ms[k] / class_n[k]), 4),
                                 "ms_per_frame": round(float(class_ms[k]), 4), "share_of_frame_kernel_time": round(float(class_ms[k] / max(frame_kernel_ms, 1e-9)), 3)}
               for k in range(8) if class_n[k] > 0}

    # one frame at a time, end to end (N = 1): what a host that calls draw() per frame sees
    latency = None
    if world == 1 and not args.no_latency:
        import ctypes as C
        reps = max(3, min(steps, 10))
        issue1 = ds.prepare_device(cfg, W, H, out.data_ptr(), 0, tile=tile, stream=stream.cuda_stream, context=0)
        for _ in range(3):
            issue1()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            issue1()
            stream.synchronize()
        dev_sync_ms = (time.perf_counter() - t0) / reps * 1e3
        lib_ = ds._lib
        cfg_abi = cfg.to_abi()
        opts = abi.RaycaRenderOptions()

        def host_frame(ptr):
            for _ in range(2):
                lib_.rayca_hip_render(ds.handle, C.byref(cfg_abi), W, H, C.byref(opts), ptr, None, None)
            t1 = time.perf_counter()
            for _ in range(reps):
                rc = lib_.rayca_hip_render(ds.handle, C.byref(cfg_abi), W, H, C.byref(opts), ptr, None, None)
                assert rc == 0, rc
            return (time.perf_counter() - t1) / reps * 1e3
        pageable = np.zeros((H, W, 4), np.uint8)
        host_ms = host_frame(pageable.ctypes.data)
        pinned = torch.empty((H, W, 4), dtype=torch.uint8, pin_memory=True)
        host_pinned_ms = host_frame(pinned.data_ptr())
        # and the rate a host WITHOUT torch reaches with frames in flight behind the C ABI: rayca_hip_render_multi_issue / _wait
        # (one device), four frame contexts in turn, every frame copied into page-locked host memory -- kernels + D2H per frame
        from rayca_amd.renderer import MultiFrames
        mf = MultiFrames([ds], cfg, W, H, band_rows=args.band_rows, gather=abi.GATHER_PEER_COPY)
        hosts = [torch.empty((H, W, 4), dtype=torch.uint8, pin_memory=True) for _ in range(4)]
        k_pipe = max(8, min(steps, 64))
        for i in range(8):
            if i >= 4:
                mf.wait(i % 4)
            mf.issue(i % 4, hosts[i % 4].data_ptr(), on_device=False)
        for c in range(4):
            mf.wait(c)
        t2 = time.perf_counter()
        for i in range(k_pipe):
            if i >= 4:
                mf.wait(i % 4)
            mf.issue(i % 4, hosts[i % 4].data_ptr(), on_device=False)
        for c in range(4):
            mf.wait(c)
        pipelined_host_ms = (time.perf_counter() - t2) / k_pipe * 1e3
        latency = {"frames_in_flight": 1,
                   "kernel_ms": round(frame_kernel_ms, 4),
                   "render_device_and_sync_ms": round(dev_sync_ms, 4),
                   "render_to_host_ms": round(host_ms, 4),
                   "render_to_pinned_host_ms": round(host_pinned_ms, 4),
                   "pipelined_to_host_ms_per_frame": round(pipelined_host_ms, 4),
                   "pipelined_to_host": "rayca_hip_render_multi_issue / _wait on one device, four frame contexts in turn, every frame copied into page-locked host memory: the frame rate a C or Rust host of the library gets, D2H included",
                   "frame_bytes_to_host": W * H * 4,
                   "note": "one frame at a time: HIP events around the frame's kernels; wall clock of rayca_hip_render_device + stream synchronise (frame stays in HBM); "
                           "wall clock of rayca_hip_render -- kernels + the copy into the caller's RGBA8 image, which is what the reference's own timer spans "
                           "(scene.rs:101-152) -- into an ordinary numpy array and into page-locked memory.  `ms_per_step` is the throughput figure with frames in flight."}

    t = torch.tensor([elapsed, elapsed_batched or 0.0, float(rays_rank)], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed_max, elapsed_batched_max, rays_total = float(tmax[0]), float(tmax[1]), float(tsum[2])
    else:
        elapsed_max, elapsed_batched_max, rays_total = elapsed, 0.0, float(rays_rank)

    ds.close()
    if rank != 0:
        return None
    mrays = rays_total * steps / elapsed_max / 1e6
    algo_rate = algo_bytes / (max(trace_frame_ms, 1e-9) * 1e-3) / 1e9
    # what MUST cross the HBM interface once per frame: the scene's arrays in (nodes, 48-B primitive records, shading
    # records), path records and pixels out -- per launch of
    # the dominant kernel: its share of the frame's traversal launches
    records = (16 + 16 + 4) * my_rows * W * max(wl_generations, 1) if is_path else 0   # direct colour, radiance factor, state per pixel and depth
    n_nodes, n_tris = int(info["node_count"]), int(info["triangle_count"])
    node_bytes = {0: (48 if node_format & 4096 else 64) * n_nodes, 1: 128 * n_nodes // 3, 4: 32 * n_nodes, 5: 64 * n_nodes // 3}[node_format & 5]
    compulsory_frame = node_bytes + (48 + 256 + 8) * n_tris + records + 4 * my_rows * W
    compulsory = compulsory_frame if launches_per_frame == 1 else (node_bytes + 48 * n_tris + (records + 4 * my_rows * W) // max(launches_per_frame, 1))
    fmt = lambda w, h: ("4-wide" if w else "binary") + (" fp16" if h else " f32")   # noqa: E731
    result = {
        "metric": "Mrays/sec (primary+shadow), 1920x1080 Sponza 1spp" if name == "atrium" else f"Mrays/sec ({name})",
        "value": round(mrays, 3),
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": round(elapsed_max / steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": wl["label"], "baseline_config": wl["baseline_config"], "width": W, "height": H, "triangles": info["triangle_count"],
                   "bvh_nodes": info["node_count"], "bvh_build_ms": round(info["build_ms"], 1), "hip_runtime_init_ms": round(info["runtime_init_ms"], 1), "bvh_built_on": "gpu (bvh_build.hip; same tree as the host builder)",
                   "node_format": {"generation0": fmt(node_format & 1, node_format & 4), "bounces": fmt(node_format & 2, node_format & 8),
                                   "binary_f32_record": "48 B: centre + half extent, child references in the low halves of two half extents (3 loads)" if node_format & 4096 else "64 B: min / max planes + references (4 loads)",
                                   "camera_rays": "lane-refill kernel (refill.hip)" if node_format & 1024 else "generation kernel",
                                   "chosen_by": "timing the four formats, then the two camera-ray kernels, on this scene (same pixels with each)"},
                   "bvh_builder": ("SAH 63 planes x 3 axes as rayca-soft bvh/blas.rs, candidate boxes seeded empty; ties by the reference's primitive order"
                                   if args.builder == "sah" else "reference SAH (rayca-soft bvh/blas.rs:64-123,261-316) incl. origin-seeded candidate boxes"),
                   "rays_per_frame": int(rays_total), "frames_in_flight": F, "tiling": f"rows in bands of {args.band_rows} dealt over {world} rank(s)",
                   "gathered_frame_equals_one_gpu_frame": frame_check,
                   "frame_gather": (f"torch.distributed.gather ({'RCCL' if backend == 'nccl' else backend}) to rank 0, ONE collective per frame at frame end, behind the frame on the frame's own stream"
                                    if world > 1 else "none"),
                   "launched_by": "torch.distributed.run / environment" if os.environ.get("TORCHELASTIC_RUN_ID") else ("bench.py (rayca_amd/launcher.py)" if world > 1 else "single process")},
        "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": None, "peak": None, "unit": "GB/s", "frac": None, "traffic": None,
                     "launch_ms": round(dom_launch_ms, 4), "launches_per_frame": round(float(class_n[dominant]), 2), "frame_kernel_ms": round(frame_kernel_ms, 4),
                     "kernels": kernels,
                     "algorithmic_access_rate": {"bytes_per_frame": int(algo_bytes), "GBps": round(algo_rate, 1), "x_hbm_peak": round(algo_rate / HBM_PEAK_GBS, 3),
                                                 "note": "SURVEY 8(d): every node and triangle a ray touches, over the frame's traversal kernels; served by L1/L2 (the tree is cache resident), so this is an access rate, not an HBM fraction",
                                                 "boxes_tested": int(counted["boxes_tested"]), "triangles_tested": int(counted["triangles_tested"]),
                                                 "hits_shaded": int(counted["hits_shaded"])},
                     "compulsory_bytes_per_launch": int(compulsory)},
    }
    if world > 1 and elapsed_batched is not None:
        result["config"]["gather_batched"] = {"frames_per_collective": B2, "ms_per_step": round(elapsed_batched_max / steps * 1e3, 4),
                                              "Mrays_per_s": round(rays_total * steps / elapsed_batched_max / 1e6, 3),
                                              "note": "the same K frames, B finished frames per collective (fewer rendezvous); reported for comparison, not as `value`"}
    rl = result["roofline"]
    counters, pmc_ms, src = pmc_for(name, args.builder, dom_name, node_format & 5, 48 if node_format & 4096 else 64) if world == 1 else (None, None, "counters describe the whole frame on one GPU")
    if counters:
        lim = roofline_limits(counters, dom_launch_ms, compulsory)
        rl.update(lim)
        rl["pmc_source"] = src
        rl["pmc_launch_ms"] = pmc_ms
        top_name, top = max(lim["limits"].items(), key=lambda kv: kv[1]["frac"])
        rl["bound"] = {"valu_issue": "valu", "l1_accesses": "l1", "l2_fill": "l2", "hbm_traffic": "hbm", "hbm_compulsory": "hbm"}[top_name]
        rl["achieved"], rl["peak"], rl["unit"], rl["frac"] = top["achieved"], top["peak"], top["unit"], top["frac"]
        rl["traffic"] = lim["hbm_traffic_bytes"]
        rl["hbm_frac"] = lim["limits"]["hbm_traffic"]["frac"]
        # the same limits over the TIMED REGION: with frames in flight the launches overlap, a frame completes every
        # ms_per_step, and the chip issues launches_per_frame x the per-launch counters in that time -- the fraction of the
        # limit the bench's operating point runs at, next to `frac` (one launch alone on the chip, with its tail)
        step_s = elapsed_max / steps
        if launches_per_frame == 1 and step_s > 0:
            rl["over_timed_region"] = {k: round(v["frac"] * (dom_launch_ms * 1e-3) / step_s, 4) for k, v in lim["limits"].items()}
            rl["over_timed_region"]["note"] = f"per-launch counters / ms_per_step ({F} frames in flight): what the chip sustains while frames overlap"
    else:   # no usable counters for this kernel / format / rank count: the compulsory-bytes HBM figure is what can be stated
        a = compulsory / (dom_launch_ms * 1e-3) / 1e9
        rl["achieved"], rl["peak"], rl["frac"] = round(a, 1), HBM_PEAK_GBS, round(a / HBM_PEAK_GBS, 4)
        rl["note"] = f"{src}: frac prices the compulsory HBM bytes only"
    if latency:
        result["latency"] = latency
    if main and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(desc, cfg, W, H, args.cpu_seconds)
    return result


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # start the ranks ourselves -- before torch or HIP are imported, so this process never owns a device
        from rayca_amd.launcher import relaunch_self
        sys.exit(relaunch_self(args.gpus))

    import torch
    import torch.distributed as dist

    env = Env()
    env.rank = rank = int(os.environ.get("RANK", "0"))
    env.world = world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    ndev = torch.cuda.device_count()
    env.dev_index = dev_index = local_rank % max(ndev, 1)   # one process per GPU; the modulo only matters for rehearsals
    torch.cuda.set_device(dev_index)
    env.dev = dev = torch.device("cuda", dev_index)
    # RCCL ("nccl") over xGMI is the real path; RAYCA_DIST_BACKEND=gloo only exists to rehearse the
    # N>1 control flow on a box with fewer GPUs than ranks (RCCL refuses two ranks on one device)
    env.backend = backend = os.environ.get("RAYCA_DIST_BACKEND", "nccl")
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
    # the streams of the frames in flight, and the comm stream, made before anything else makes streams: each on a
    # hardware queue of its own (rayca_amd/streams.py -- two frame streams on one queue cost a rank's eighth 0.12-0.16 ms
    # per frame instead of 0.073)
    from rayca_amd.streams import frame_streams
    env.frame_streams, (env.comm,) = frame_streams(dev, 8, spare=1)

    result = run_workload(env, args, args.workload, args.steps, args.warmup, main=True)
    others = []
    if args.workload == "atrium" and not args.no_others:
        for name, k, w in OTHER_WORKLOADS:
            r = run_workload(env, args, name, k, w, main=False)
            if r is not None:
                others.append({"workload": r["config"]["workload"], "name": name, "baseline_config": r["config"]["baseline_config"],
                               "Mrays_per_s": r["value"], "ms_per_step": r["ms_per_step"], "steps": r["steps"], "warmup": r["warmup"],
                               "rays_per_frame": r["config"]["rays_per_frame"], "frames_in_flight": r["config"]["frames_in_flight"], "gathered_frame_equals_one_gpu_frame": r["config"]["gathered_frame_equals_one_gpu_frame"],
                               "triangles": r["config"]["triangles"], "bvh_build_ms": r["config"]["bvh_build_ms"], "node_format": r["config"]["node_format"],
                               "roofline": r["roofline"], **({"latency": r["latency"]} if "latency" in r else {})})
    if rank == 0:
        if others:
            result["config"]["other_workloads"] = others
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
                      f"{st['seconds']:.2f} s wall); per-test vertex transforms as in the reference, on the reference's own "
                      "(origin-seeded) tree; BVH build excluded like the reference's own timer (scene.rs:101,152)"}


if __name__ == "__main__":
    main()
