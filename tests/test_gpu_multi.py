"""rayca_hip_render_multi: several devices behind the C ABI.  The one-GPU box can rehearse everything but
the RCCL exchange itself: several scene handles on ONE device with the peer-copy transport exercise the row split, the
per-device threads, the gather layout and the de-interleave kernel; RCCL's own path needs distinct devices."""
import numpy as np
import pytest

from rayca_amd import Config, DeviceScene, IntegratorStrategy, abi, flatten, lib, scenes
from rayca_amd.lib import RaycaError
from rayca_amd.renderer import render_multi


def test_rccl_can_be_opened(product_lib):
    """no GPU needed: the library opens librccl at first use and reports what it found"""
    rc = product_lib.rayca_hip_rccl_status()
    assert rc == abi.OK, lib.last_error()


@pytest.mark.gpu
@pytest.mark.parametrize("parts,size,band", [(1, (320, 180), 8), (2, (320, 181), 8), (3, (257, 99), 4), (8, (640, 360), 8), (5, (64, 7), 8)])
def test_multi_equals_single(gpu, parts, size, band):
    w, h = size
    desc = flatten(scenes.cornell_scene())
    handles = [DeviceScene(desc, Config(), builder=abi.BUILDER_SAH) for _ in range(parts)]
    for cfg in (Config(integrator=IntegratorStrategy.Flat), Config(max_depth=2, seed=3)):
        want, _, st1 = handles[0].render(cfg, w, h, want_f32=False, collect_stats=True)
        got, st = render_multi(handles, cfg, w, h, band_rows=band, gather=abi.GATHER_PEER_COPY, collect_stats=True)
        assert np.array_equal(got, want)
        assert sum(s["rows_rendered"] for s in st) == h                                 # (a part with no rows reports 0)
        for k in ("rays_primary", "rays_shadow", "rays_bounce", "hits_shaded"):
            assert sum(s[k] for s in st) == st1[k], k
        again, _ = render_multi(handles, cfg, w, h, band_rows=band, gather=abi.GATHER_PEER_COPY)     # state is reused
        assert np.array_equal(again, want)
    for hd in handles:
        hd.close()


@pytest.mark.gpu
def test_multi_argument_errors(gpu):
    desc = flatten(scenes.box_scene())
    a, b = DeviceScene(desc, Config()), DeviceScene(desc, Config())
    cfg = Config(integrator=IntegratorStrategy.Flat)
    with pytest.raises(RaycaError) as e:
        render_multi([a, b], cfg, 64, 64, gather=abi.GATHER_RCCL)       # two parts on one device: RCCL refuses that, so do we
    assert e.value.code == abi.ERR_BAD_ARG
    with pytest.raises(RaycaError) as e:
        render_multi([a, a], cfg, 64, 64, gather=abi.GATHER_PEER_COPY)
    assert e.value.code == abi.ERR_BAD_ARG
    with pytest.raises(RaycaError) as e:
        render_multi([a, b], cfg, 64, 64, gather=7)
    assert e.value.code == abi.ERR_BAD_ARG
    one, _ = render_multi([a], cfg, 64, 64, gather=abi.GATHER_RCCL)       # a single part needs no exchange with either transport
    assert np.array_equal(one, a.render(cfg, 64, 64, want_f32=False)[0])


@pytest.mark.gpu
@pytest.mark.parametrize("parts", [1, 3])
def test_frames_in_flight_through_the_multi_entry(gpu, parts):
    """rayca_hip_render_multi_issue / _wait: frames of DIFFERENT Configs issued back to back on different frame contexts
    (nothing waited for in between), then waited for -- each must be the frame rayca_hip_render gives for its Config, bit
    for bit; a context is re-used after its wait; frames of one context are serialised."""
    from rayca_amd.renderer import MultiFrames
    w, h = 320, 184
    desc = flatten(scenes.cornell_scene())
    handles = [DeviceScene(desc, Config(), builder=abi.BUILDER_SAH) for _ in range(parts)]
    cfgs = [Config(integrator=IntegratorStrategy.Flat), Config(max_depth=1), Config(max_depth=2, seed=3), Config(max_depth=3, seed=5)]
    want = [handles[0].render(c, w, h, want_f32=False)[0] for c in cfgs]
    loops = [MultiFrames(handles, c, w, h, band_rows=8, gather=abi.GATHER_PEER_COPY) for c in cfgs]
    for rounds in range(3):
        outs = [np.zeros((h, w, 4), np.uint8) for _ in cfgs]
        for ctx, (loop, out) in enumerate(zip(loops, outs)):
            loop.issue(ctx, out)                       # four frames in flight, one per context
        for ctx, loop in enumerate(loops):
            loop.wait(ctx)
        for got, ref in zip(outs, want):
            assert np.array_equal(got, ref)
    # the same context twice in a row without a wait in between: the second frame is queued behind the first
    a, b = np.zeros((h, w, 4), np.uint8), np.zeros((h, w, 4), np.uint8)
    loops[1].issue(5, a)
    loops[2].issue(5, b)
    loops[2].wait(5)
    assert np.array_equal(b, want[2])
    assert np.array_equal(a, want[1])     # (its copy-out precedes the second frame's on the same stream)
    loops[0].wait(7)                      # a context nothing was issued on: returns at once
    # the synchronous entry on a context of its own, next to them
    got, _ = render_multi(handles, cfgs[3], w, h, band_rows=8, gather=abi.GATHER_PEER_COPY, context=6)
    assert np.array_equal(got, want[3])
    with pytest.raises(RaycaError) as e:
        render_multi(handles, cfgs[0], w, h, gather=abi.GATHER_PEER_COPY, context=8)
    assert e.value.code == abi.ERR_BAD_ARG
    for hd in handles:
        hd.close()


@pytest.mark.gpu
def test_render_device_waits_for_and_records_the_callers_events(gpu):
    """RaycaRenderOptions.wait_event / record_event: one native call per frame of a frame loop.  The frame must not start
    before the event it waits for (recorded behind a long fill on another stream that writes the target buffer), and the
    recorded event must cover the frame (a copy on a third stream that waits for it sees the finished pixels)."""
    import torch
    w, h = 256, 144
    cfg = Config(max_depth=1)
    ds = DeviceScene(flatten(scenes.cornell_scene()), cfg, builder=abi.BUILDER_SAH)
    want = ds.render(cfg, w, h, want_f32=False)[0]
    dev = torch.device("cuda", 0)
    s_fill, s_frame, s_copy = (torch.cuda.Stream(dev) for _ in range(3))
    out = torch.zeros((h, w, 4), dtype=torch.uint8, device=dev)
    big = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
    ev_filled, ev_frame = torch.cuda.Event(), torch.cuda.Event()
    # torch creates an event's handle at its first record: give both one, so that their handles can be passed
    ev_filled.record(s_fill)
    ev_frame.record(s_frame)
    torch.cuda.synchronize()
    issue = ds.prepare_device(cfg, w, h, out.data_ptr(), 0, stream=s_frame.cuda_stream, wait_event=ev_filled.cuda_event, record_event=ev_frame.cuda_event)
    for _ in range(3):
        with torch.cuda.stream(s_fill):
            for _ in range(4):
                big.fill_(7)          # ~1 GB of writes in front of ...
            out.fill_(9)              # ... the write the frame has to come after
            ev_filled.record(s_fill)
        issue()
        with torch.cuda.stream(s_copy):
            s_copy.wait_event(ev_frame)
            got = out.clone()
        torch.cuda.synchronize()
        assert np.array_equal(got.cpu().numpy(), want)
    ds.close()
