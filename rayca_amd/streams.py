"""Streams for frames in flight that sit on different hardware queues.

HIP (ROCm 7) runs all streams of a process over four hardware queues per device (GPU_MAX_HW_QUEUES).  A stream is tied to
one of them when it is CREATED: the queue with the fewest streams on it at that moment, the first such in the runtime's own
order (AMD_LOG_LEVEL=4: "Selected queue refCount", profiles/r02_queue_log.txt).  Two frame streams that end up on one queue
run their kernels one after the other, and a rank's share of a frame then takes 0.12-0.16 ms instead of 0.073
(tests/gpu_rank_share_probe.py) -- and whether they do depends on which other streams (the library's, RCCL's, torch's) happen
to exist when they are made.

Streams made back to back level the queues out first and then go round them in turn, so of a run of consecutive creations
the LAST four are on four different queues whatever the state before (as long as nobody else makes or drops a stream in
between).  torch makes the 32 streams of its pool in one go at the first request and hands them out in creation order:
asking for sixteen and keeping the last ones is that run."""
from __future__ import annotations

import torch


def frame_streams(device, count: int, *, spare: int = 1, run: int = 16):
    """`count` streams for frames in flight plus `spare` more (a comm stream), taken from the end of a run of `run` pool streams.
    Call it before other threads of the process start making streams of their own (the scene's background thread does for a
    few milliseconds after rayca_hip_scene_create: make the streams first, or after rayca_hip_scene_finish)."""
    if count + spare > run:
        raise ValueError("more streams asked for than the run holds")
    pool = [torch.cuda.Stream(device) for _ in range(run)]
    for s in pool:   # (the pool exists once its first stream is asked for; touching each costs nothing and keeps torch honest)
        s.synchronize()
    frames = pool[run - spare - count: run - spare]
    spares = pool[run - spare:]
    return frames, spares
