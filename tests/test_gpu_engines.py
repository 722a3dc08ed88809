"""The two generation-by-generation engines -- the fused persistent kernel (k_generation) and the wavefront split
(lean one-thread-per-ray trace kernels + streaming shade kernel, wavefront.inc) -- must produce the same bits:
they share every arithmetic routine and differ only in scheduling."""
import numpy as np
import pytest

import oracle_lib as ol
from parity_report import check_outliers
from rayca_amd import Config, DeviceScene, IntegratorStrategy, SamplerStrategy, abi, flatten, scenes
import test_gpu_general as G

pytestmark = pytest.mark.gpu
I, S = IntegratorStrategy, SamplerStrategy


def both(ds, cfg, w, h, **kw):
    a = ds.render(cfg, w, h, engine=abi.ENGINE_FUSED, collect_stats=True, **kw)
    b = ds.render(cfg, w, h, engine=abi.ENGINE_WAVEFRONT, collect_stats=True, **kw)
    return a, b


CASES = [
    ("cornell", Config(integrator=I.Flat), 320, 180),
    ("cornell", Config(max_depth=1), 321, 179),                       # ragged: not a multiple of the 8x8 tile
    ("cornell", Config(max_depth=4, seed=3), 320, 180),
    ("cornell", Config(max_depth=3, indirect_sampler=S.Hemisphere, samples_per_pixel=4, gamma=2.2, seed=4), 160, 90),
    ("room_phong", Config(max_depth=3, direct_sampler=S.NONE, seed=5), 160, 90),
    ("room_phong", Config(max_depth=1, light_samples=4, light_stratify=True, seed=6), 160, 120),
    ("room_ggx", Config(max_depth=3, seed=7), 160, 120),
    ("box", Config(), 200, 200),
    ("box", Config(integrator=I.Flat, samples_per_pixel=4), 64, 64),
    ("sphere", Config(max_depth=2, seed=11), 128, 128),               # a scaled sphere (model-space hit ray, sphere.rs:138)
    ("sphere_boxes", Config(max_depth=3, seed=12), 160, 120),
    ("textured", Config(max_depth=2, seed=13), 128, 128),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_engines_agree(gpu, case):
    name, cfg, w, h = CASES[case]
    ds, _ = G.pair(name)
    (u8a, fa, sa), (u8b, fb, sb) = both(ds, cfg, w, h)
    assert np.array_equal(fa.view(np.uint32), fb.view(np.uint32)), f"max abs diff {np.abs(fa - fb).max():.3e}"
    assert np.array_equal(u8a, u8b)
    for k in ("rays_primary", "rays_shadow", "rays_bounce", "hits_shaded"):
        assert sa[k] == sb[k], k
    assert float(fa[..., :3].max()) > 0.0
    # RAYCA_ENGINE_AUTO picks one of the two by the number of generations: always the same frame and the same ray counts
    u8c, fc, sc = ds.render(cfg, w, h, engine=abi.ENGINE_AUTO, collect_stats=True)
    assert np.array_equal(fc.view(np.uint32), fa.view(np.uint32)) and np.array_equal(u8c, u8a)
    for k in ("rays_primary", "rays_shadow", "rays_bounce", "hits_shaded", "boxes_tested", "triangles_tested"):
        assert sc[k] == sa[k] or k in ("boxes_tested", "triangles_tested"), k
    # SIMD-slot accounting (include/rayca_hip.h): a wave books 64 lanes per trip through the node / leaf loop, so the
    # slots bound the per-lane tests from above (the root box of every ray is tested outside the node loop)
    rays = sa["rays_primary"] + sa["rays_shadow"] + sa["rays_bounce"]
    for st in (sa, sb):
        assert st["wave_box_slots"] >= st["boxes_tested"] - rays >= 0   # == 0: the root is a leaf (single sphere)
        assert st["wave_triangle_slots"] >= st["triangles_tested"] > 0


def test_engines_agree_on_tiles_and_builders(gpu):
    desc = flatten(scenes.cornell_scene())
    cfg = Config(max_depth=2, seed=9)
    ref = None
    for builder in (abi.BUILDER_REFERENCE, abi.BUILDER_SAH):
        ds = DeviceScene(desc, Config(), builder=builder)
        for engine in (abi.ENGINE_FUSED, abi.ENGINE_WAVEFRONT):
            frame = np.zeros((135, 240, 4), np.float32)
            for part in range(3):
                _, f32, _ = ds.render(cfg, 240, 135, tile=(part, 3, 8), engine=engine)
                rows = [y for y in range(135) if (y // 8) % 3 == part]
                frame[rows] = f32
            if ref is None:
                ref = frame
            assert np.array_equal(ref.view(np.uint32), frame.view(np.uint32)), (builder, engine)


def test_engines_agree_at_full_size(gpu):
    """The bench frame (atrium 1080p, primary + shadow) and the 4-bounce frame at reduced size."""
    ds = DeviceScene(flatten(scenes.atrium_scene()), Config(), builder=abi.BUILDER_SAH)
    for cfg, w, h in ((Config(max_depth=1), 1920, 1080), (Config(integrator=I.Flat), 1920, 1080), (Config(seed=2), 640, 360)):
        (u8a, fa, sa), (u8b, fb, sb) = both(ds, cfg, w, h, want_f32=True)
        assert np.array_equal(fa.view(np.uint32), fb.view(np.uint32)), f"max abs diff {np.abs(fa - fb).max():.3e}"
        assert sa["rays_shadow"] == sb["rays_shadow"] and sa["rays_bounce"] == sb["rays_bounce"]


def test_config4_4k_four_bounce_engines_and_eight_tiles(gpu):
    """BASELINE configs[4] at full size: the atrium at 3840x2160, 4-bounce path trace.  Far beyond what the oracle can
    render, so checked through properties: the two engines give the same bits, and the frame assembled from the
    eight row tiles the eight ranks would render equals the frame rendered whole (pixels depend on nothing but
    their own coordinates and the seed)."""
    w, h = 3840, 2160
    cfg = Config(max_depth=5, seed=4)
    ds = DeviceScene(flatten(scenes.atrium_scene()), Config(), builder=abi.BUILDER_SAH)
    whole, _, st = ds.render(cfg, w, h, want_f32=False, engine=abi.ENGINE_WAVEFRONT)
    fused, _, sf = ds.render(cfg, w, h, want_f32=False, engine=abi.ENGINE_FUSED)
    assert np.array_equal(whole, fused)
    assert st["rays_primary"] == w * h and st["rays_shadow"] == sf["rays_shadow"] > w * h and st["rays_bounce"] == sf["rays_bounce"] > w * h
    frame = np.zeros_like(whole)
    shadow = bounce = 0
    for part in range(8):
        rows = [y for y in range(h) if (y // 8) % 8 == part]
        u8, _, sp = ds.render(cfg, w, h, tile=(part, 8, 8), want_f32=False)
        assert u8.shape[0] == len(rows) == sp["rows_rendered"]
        frame[rows] = u8
        shadow += sp["rays_shadow"]
        bounce += sp["rays_bounce"]
    assert np.array_equal(frame, whole)
    assert shadow == st["rays_shadow"] and bounce == st["rays_bounce"]
    assert int(whole[..., :3].max()) > 0


def test_frames_in_flight_on_separate_contexts(gpu):
    """Three frames of one scene in flight on three streams (RaycaRenderOptions.context): each must equal the frame
    rendered alone.  Different configs per context, so a shared work buffer would show."""
    import torch
    ds, _ = G.pair("cornell")
    dev = torch.device("cuda", 0)
    cfgs = [Config(max_depth=3, seed=1), Config(max_depth=1), Config(integrator=I.Flat)]
    w, h = 640, 360
    alone = [ds.render(c, w, h, want_f32=False)[0] for c in cfgs]
    streams = [torch.cuda.Stream(dev) for _ in cfgs]
    outs = [torch.zeros((h, w, 4), dtype=torch.uint8, device=dev) for _ in cfgs]
    for _ in range(5):
        for i, c in enumerate(cfgs):
            ds.render_device(c, w, h, outs[i].data_ptr(), 0, stream=streams[i].cuda_stream, context=i)
    torch.cuda.synchronize()
    for i in range(len(cfgs)):
        assert np.array_equal(outs[i].cpu().numpy(), alone[i]), i
    with pytest.raises(Exception):
        ds.render(cfgs[0], 8, 8, context=9)


def test_stack_machine_on_tiles_and_with_samples(gpu):
    """The per-pixel stack machine under the multi-GPU row split and with several samples per pixel."""
    ds, orc = G.pair("room_phong")
    cfg = Config(integrator=I.Direct, light_samples=2, samples_per_pixel=4, gamma=2.2, seed=31)
    w, h = 96, 75   # ragged: 75 rows in bands of 8 over 3 parts
    _, full, _ = ds.render(cfg, w, h)
    _, ofull, _ = orc.render(cfg, w, h)
    check_outliers("room_phong_direct_spp4_96x75", full, ofull)
    frame = np.zeros_like(full)
    for part in range(3):
        _, f32, st = ds.render(cfg, w, h, tile=(part, 3, 8), collect_stats=True)
        rows = [y for y in range(h) if (y // 8) % 3 == part]
        assert st["rows_rendered"] == len(rows)
        frame[rows] = f32
    assert np.array_equal(frame.view(np.uint32), full.view(np.uint32))


# ---- lane-refill camera-ray kernel (rayca_amd/csrc/refill.hip) ------------------------------------------------------
REFILL_SCENES = {
    "box": (scenes.box_scene, [(256, 256), (65, 64), (33, 17), (1, 1)]),
    "cornell": (scenes.cornell_scene, [(321, 179), (640, 360)]),
    "sphere": (lambda: G.sphere_scenes(False), [(128, 128)]),
    "sphere_boxes": (lambda: G.sphere_scenes(True), [(160, 120)]),
    "textured": (G.textured_scene, [(200, 200)]),
    "soup20k": (lambda: scenes.soup_scene(20000, extent=0.03), [(384, 384)]),
    "atrium3": (lambda: scenes.atrium_scene(detail=3), [(320, 180)]),
}


@pytest.mark.parametrize("name", list(REFILL_SCENES))
def test_lane_refill_kernel_equals_the_generation_kernel_and_the_oracle(gpu, name):
    """Flat frames: a wave that refills finished lanes with new pixels must give every pixel the bits the fused generation
    kernel gives it -- and the oracle's.  Per-ray work is the same too (same steps in the same order): equal box and
    triangle test counts; only the SIMD-slot accounting differs (that is the point)."""
    make, sizes = REFILL_SCENES[name]
    desc = flatten(make())
    ds = DeviceScene(desc, Config(), builder=abi.BUILDER_SAH)
    orc = ol.OracleScene(desc, Config(), build=ol.BUILD_BINNED)
    flat = Config(integrator=I.Flat)
    for (w, h) in sizes:
        ua, fa, sa = ds.render(flat, w, h, camera_rays=abi.CAMERA_GENERATION, collect_stats=True)
        ub, fb, sb = ds.render(flat, w, h, camera_rays=abi.CAMERA_REFILL, collect_stats=True)
        assert not sa["node_format"] & 1024 and sb["node_format"] & 1024
        assert np.array_equal(fa.view(np.uint32), fb.view(np.uint32)) and np.array_equal(ua, ub)
        for k in ("rays_primary", "hits_shaded", "boxes_tested", "triangles_tested"):
            assert sa[k] == sb[k], k
        ou, of, _ = orc.render(flat, w, h)
        assert np.array_equal(fb.view(np.uint32), of.view(np.uint32)) and np.array_equal(ub, ou)
    # under the multi-GPU row split (ragged last band)
    w, h = sizes[0]
    if h >= 17:
        _, whole, _ = ds.render(flat, w, h, camera_rays=abi.CAMERA_REFILL)
        frame = np.zeros_like(whole)
        for part in range(3):
            _, f32, st = ds.render(flat, w, h, tile=(part, 3, 8), camera_rays=abi.CAMERA_REFILL)
            rows = [y for y in range(h) if (y // 8) % 3 == part]
            assert st["rows_rendered"] == len(rows)
            frame[rows] = f32
        assert np.array_equal(frame.view(np.uint32), whole.view(np.uint32))
    ds.close()
    orc.close()


def test_lane_refill_at_full_size_and_its_lane_utilisation(gpu):
    """The atrium's 1080p camera rays on both kernels: same frame, same per-ray work.  (Lane utilisation is printed, not
    asserted: on coherent rays refilling LOWERS it -- new rays descending from the root hold up the lanes that already
    have a leaf -- which is why the kernel is a per-scene choice made by timing.)"""
    ds = DeviceScene(flatten(scenes.atrium_scene()), Config(), builder=abi.BUILDER_SAH)
    flat = Config(integrator=I.Flat)
    ua, _, sa = ds.render(flat, 1920, 1080, camera_rays=abi.CAMERA_GENERATION, collect_stats=True, want_f32=False)
    ub, _, sb = ds.render(flat, 1920, 1080, camera_rays=abi.CAMERA_REFILL, collect_stats=True, want_f32=False)
    assert np.array_equal(ua, ub)
    assert sa["boxes_tested"] == sb["boxes_tested"] and sa["triangles_tested"] == sb["triangles_tested"]
    util_a, util_b = sa["boxes_tested"] / sa["wave_box_slots"], sb["boxes_tested"] / sb["wave_box_slots"]
    print(f"[refill] atrium 1080p node-loop lane utilisation: generation {util_a:.3f}, refill {util_b:.3f}")
    with pytest.raises(Exception):
        ds.render(flat, 64, 64, camera_rays=7)
    ds.close()


def test_scene_lifetime_around_the_background_formats(gpu):
    """scene_create leaves the 4-wide / fp16 node formats to a thread of the scene's own.  A scene may be destroyed at any
    point of that thread's life (it stops at its next phase boundary), frames issued before the formats are there run on
    the binary nodes, and the frame is the same before and after."""
    desc = flatten(scenes.atrium_scene())
    cfg = Config(max_depth=1)
    DeviceScene(desc, Config(), builder=abi.BUILDER_SAH).close()            # destroyed at once, not a frame rendered
    ds = DeviceScene(desc, Config(), builder=abi.BUILDER_SAH)
    early, _, st = ds.render(cfg, 640, 360, want_f32=False, collect_stats=True)
    ds.close()                                                               # destroyed while the thread may still be running
    ds = DeviceScene(desc, Config(), builder=abi.BUILDER_SAH)
    ds.finish()
    info = ds.info()
    assert info["node_count"] > 0 and info["triangle_count"] == 271568
    late, _, st2 = ds.render(cfg, 640, 360, want_f32=False, collect_stats=True)
    assert not st2["node_format"] & 2048
    assert np.array_equal(early, late)
    assert st["boxes_tested"] > 0 and st2["triangles_tested"] > 0
    ds.finish()                                                              # a second wait is a no-op
    ds.close()


def test_frame_streams_helper(gpu):
    """rayca_amd.streams.frame_streams: the last streams of one run of pool streams, all different, and frames issued on them
    with one frame context each come out as the frame rendered alone."""
    import torch
    from rayca_amd.streams import frame_streams
    dev = torch.device("cuda", 0)
    frames, spares = frame_streams(dev, 4, spare=1)
    handles = [s.cuda_stream for s in frames + spares]
    assert len(frames) == 4 and len(spares) == 1 and len(set(handles)) == 5
    with pytest.raises(ValueError):
        frame_streams(dev, 16, spare=1)
    ds = DeviceScene(flatten(scenes.cornell_scene()), Config())
    cfg = Config(max_depth=2)
    alone, _, _ = ds.render(cfg, 320, 180, want_f32=False)
    outs = [torch.zeros((180, 320, 4), dtype=torch.uint8, device=dev) for _ in frames]
    for rnd in range(3):
        for i, s in enumerate(frames):
            ds.render_device(cfg, 320, 180, outs[i].data_ptr(), 0, stream=s.cuda_stream, context=i)
    torch.cuda.synchronize()
    for o in outs:
        assert np.array_equal(o.cpu().numpy(), alone)
    ds.close()
