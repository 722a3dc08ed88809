"""Mirror of rayca-soft's renderer surface on top of the C ABI: `Config` (config.rs:10-49),
`IntegratorStrategy`, `SamplerStrategy`, `SoftRenderer` with `draw(scene, image)` (scene.rs:88-154).
All compute happens in librayca_hip.so on the GPU."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import abi, lib
from .model import Image, Scene, create_default_model, flatten


class IntegratorStrategy:
    Scratcher, Raytracer, Flat, AnalyticDirect, Direct, Pathtracer = range(6)


class SamplerStrategy:
    NONE, Nee, Hemisphere, Cosine, Brdf, Mis = range(6)


@dataclass
class Config:
    """rayca_soft::Config with the reference's defaults (config.rs:10-49)."""
    bvh: bool = True
    light_samples: int = 1
    light_stratify: bool = False
    samples_per_pixel: int = 1
    russian_roulette: bool = False
    direct_sampler: int = SamplerStrategy.Nee
    indirect_sampler: int = SamplerStrategy.Cosine
    integrator: int = IntegratorStrategy.Pathtracer
    max_depth: int = 5
    gamma: float = 1.0
    seed: int = 0

    def to_abi(self) -> abi.RaycaConfig:
        c = abi.RaycaConfig()
        c.bvh, c.light_samples, c.light_stratify = int(self.bvh), self.light_samples, int(self.light_stratify)
        c.samples_per_pixel, c.russian_roulette = self.samples_per_pixel, int(self.russian_roulette)
        c.direct_sampler, c.indirect_sampler, c.integrator = self.direct_sampler, self.indirect_sampler, self.integrator
        c.max_depth, c.gamma, c.seed = self.max_depth, self.gamma, self.seed
        return c


class DeviceScene:
    """Owns a RaycaScene handle: the device-resident scene + BVH (first half of draw, scene.rs:90-99)."""

    def __init__(self, desc: abi.SceneDesc, config: Optional[Config] = None, device: int = 0,
                 builder: int = abi.BUILDER_REFERENCE, build_on_host: bool = False, _lib=None):
        self._lib = _lib or lib.load()  # _lib: an explicitly loaded build of the library (A/B experiments)
        self.desc = desc
        cfg = (config or Config()).to_abi()
        opts = abi.RaycaBuildOptions()
        opts.builder, opts.device, opts.build_on_host = builder, device, int(build_on_host)
        h = C.c_void_p()
        lib.check(self._lib.rayca_hip_scene_create(desc.ptr(), C.byref(cfg), C.byref(opts), C.byref(h)))
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self._lib.rayca_hip_scene_destroy(self.handle)
            self.handle = None

    __del__ = close

    def info(self) -> dict:
        i = abi.RaycaSceneInfo()
        lib.check(self._lib.rayca_hip_scene_info(self.handle, C.byref(i)))
        return i.as_dict()

    def finish(self) -> None:
        """Wait for the node formats scene_create left to its own thread (rayca_hip_scene_finish): only needed where the
        first frames must already be eligible for every format (format calibration tests, benchmarks)."""
        lib.check(self._lib.rayca_hip_scene_finish(self.handle))

    def primitive_order(self) -> np.ndarray:
        n = self.info()["triangle_count"] + self.info()["sphere_count"]
        out = np.zeros(n, np.uint32)
        lib.check(self._lib.rayca_hip_scene_primitive_order(self.handle, out.ctypes.data, n))
        return out

    def read_nodes(self, which: int) -> np.ndarray:
        """rayca_hip_scene_read_nodes: which = 0 the 64-B binary nodes as (N, 16) uint32, 1 the 48-B centre / half records as (N, 12)"""
        n = C.c_uint64(0)
        lib.check(self._lib.rayca_hip_scene_read_nodes(self.handle, which, None, 0, C.byref(n)))
        out = np.zeros(n.value // 4, np.uint32)
        lib.check(self._lib.rayca_hip_scene_read_nodes(self.handle, which, out.ctypes.data, n.value, None))
        return out.reshape(-1, 16 if which == 0 else 12)

    @staticmethod
    def _opts(traversal, collect_stats, tile, stream, engine=abi.ENGINE_AUTO, context=0, camera_rays=abi.CAMERA_AUTO):
        o = abi.RaycaRenderOptions()
        o.traversal, o.collect_stats = traversal, int(collect_stats)
        o.engine, o.context, o.camera_rays = engine, context, camera_rays
        if tile is not None:
            o.tile.part, o.tile.parts, o.tile.band_rows = tile
        o.stream = stream
        return o

    def tile_rows(self, tile, height) -> int:
        t = abi.RaycaTile()
        t.part, t.parts, t.band_rows = tile
        return self._lib.rayca_hip_tile_rows(C.byref(t), height)

    def render(self, config: Config, width: int, height: int, *, traversal=abi.TRAVERSAL_ORDERED,
               collect_stats=False, tile=None, want_rgba8=True, want_f32=True, engine=abi.ENGINE_AUTO, context=0,
               camera_rays=abi.CAMERA_AUTO):
        """rayca_hip_render: host outputs. Returns (rgba8 | None, rgba32f | None, stats dict)."""
        rows = height if tile is None else self.tile_rows(tile, height)
        u8 = np.zeros((rows, width, 4), np.uint8) if want_rgba8 else None
        f32 = np.zeros((rows, width, 4), np.float32) if want_f32 else None
        st = abi.RaycaStats()
        cfg = config.to_abi()
        o = self._opts(traversal, collect_stats, tile, None, engine, context, camera_rays)
        lib.check(self._lib.rayca_hip_render(self.handle, C.byref(cfg), width, height, C.byref(o),
                                             u8.ctypes.data if u8 is not None else None,
                                             f32.ctypes.data if f32 is not None else None, C.byref(st)))
        return u8, f32, st.as_dict()

    def render_device(self, config: Config, width: int, height: int, d_rgba8: int, d_f32: int = 0, *,
                      traversal=abi.TRAVERSAL_ORDERED, collect_stats=False, tile=None, stream=None,
                      want_stats=False, engine=abi.ENGINE_AUTO, context=0, camera_rays=abi.CAMERA_AUTO):
        """rayca_hip_render_device: outputs stay in device memory (pointers as ints)."""
        st = abi.RaycaStats()
        cfg = config.to_abi()
        o = self._opts(traversal, collect_stats, tile, stream, engine, context, camera_rays)
        lib.check(self._lib.rayca_hip_render_device(self.handle, C.byref(cfg), width, height, C.byref(o),
                                                    d_rgba8 or None, d_f32 or None,
                                                    C.byref(st) if want_stats else None))
        return st.as_dict() if want_stats else None

    def prepare_device(self, config: Config, width: int, height: int, d_rgba8: int, d_f32: int = 0, *,
                       traversal=abi.TRAVERSAL_ORDERED, tile=None, stream=None, engine=abi.ENGINE_AUTO, context=0,
                       camera_rays=abi.CAMERA_AUTO, wait_event=None, record_event=None):
        """rayca_hip_render_device with every argument marshalled once: returns a zero-argument callable that issues the
        frame (asynchronously, no statistics).  A frame loop that renders the same frame into the same buffer again and
        again -- bench.py's ranks, a viewer -- pays the ctypes marshalling once instead of per frame (~10 us of the
        ~50 us a rank's share of a 1080p frame costs on the host)."""
        cfg = config.to_abi()
        o = self._opts(traversal, False, tile, stream, engine, context, camera_rays)
        # hipEvent_t handles (ints): the frame's stream waits for the one before its first kernel and records the other
        # behind its last -- wait + render + record in ONE native call (RaycaRenderOptions.wait_event / record_event)
        o.wait_event, o.record_event = wait_event, record_event
        fn, handle, check = self._lib.rayca_hip_render_device, self.handle, lib.check
        args = (handle, C.byref(cfg), width, height, C.byref(o), d_rgba8 or None, d_f32 or None, None)

        def issue(_keep=(cfg, o)):
            rc = fn(*args)
            if rc:
                check(rc)
        return issue

    def trace_rays(self, rays: np.ndarray, *, traversal=abi.TRAVERSAL_ORDERED, collect_stats=False):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        n = rays.shape[0]
        t = np.zeros(n, np.float32)
        prim = np.zeros(n, np.uint32)
        uv = np.zeros((n, 2), np.float32)
        st = abi.RaycaStats()
        o = self._opts(traversal, collect_stats, None, None)
        lib.check(self._lib.rayca_hip_trace_rays(self.handle, C.byref(o), n, rays.ctypes.data, t.ctypes.data,
                                                 prim.ctypes.data, uv.ctypes.data, C.byref(st)))
        return t, prim, uv, st.as_dict()


def _multi_args(scenes, config, band_rows, gather, traversal, collect_stats, engine, context):
    handles = (C.c_void_p * len(scenes))(*[s.handle for s in scenes])
    o = abi.RaycaMultiOptions()
    o.traversal, o.collect_stats, o.band_rows, o.gather, o.engine, o.context = traversal, int(collect_stats), band_rows, gather, engine, context
    return handles, o, config.to_abi()


def render_multi(scenes, config: Config, width: int, height: int, *, band_rows=8, gather=abi.GATHER_RCCL, traversal=abi.TRAVERSAL_ORDERED,
                 collect_stats=False, engine=abi.ENGINE_AUTO, want_stats=False, context=0):
    """rayca_hip_render_multi: one frame on len(scenes) devices of this process, scenes[i] = the DeviceScene that renders
    part i (created on its own device); returns (rgba8 (H, W, 4), [stats per device] | None)."""
    l = scenes[0]._lib
    handles, o, cfg = _multi_args(scenes, config, band_rows, gather, traversal, collect_stats, engine, context)
    u8 = np.zeros((height, width, 4), np.uint8)
    st = (abi.RaycaStats * len(scenes))() if (want_stats or collect_stats) else None
    lib.check(l.rayca_hip_render_multi(handles, len(scenes), C.byref(cfg), width, height, C.byref(o), u8.ctypes.data, st))
    return u8, ([x.as_dict() for x in st] if st is not None else None)


class MultiFrames:
    """Frames in flight through the several-devices entry (rayca_hip_render_multi_issue / _wait): `issue(context, out)`
    queues a whole frame -- every part's kernels, the one exchange, de-interleave, copy into `out` -- on frame context
    `context` of every scene and returns; `wait(context)` blocks until that frame has landed.  A host that walks the
    contexts in turn keeps that many frames in flight on every device (what a C or Rust host of the library does)."""

    def __init__(self, scenes, config: Config, width: int, height: int, *, band_rows=8, gather=abi.GATHER_RCCL,
                 traversal=abi.TRAVERSAL_ORDERED, engine=abi.ENGINE_AUTO):
        self.scenes, self.width, self.height = list(scenes), width, height
        self._l = scenes[0]._lib
        self._args = {}
        self._mk = lambda ctx: _multi_args(self.scenes, config, band_rows, gather, traversal, False, engine, ctx)

    def issue(self, context: int, out, on_device=None) -> None:
        """out: a (H, W, 4) uint8 numpy array (kept alive by the caller until wait), or a pointer (int): device memory of
        scenes[0]'s device unless on_device=False (then host memory, e.g. a page-locked torch tensor's data_ptr)."""
        if context not in self._args:
            self._args[context] = self._mk(context)
        handles, o, cfg = self._args[context]
        if on_device is None:
            on_device = isinstance(out, int)
        o.output_on_device = int(bool(on_device))
        ptr = out if isinstance(out, int) else out.ctypes.data
        lib.check(self._l.rayca_hip_render_multi_issue(handles, len(self.scenes), C.byref(cfg), self.width, self.height, C.byref(o), ptr))

    def wait(self, context: int) -> None:
        handles = self._args[context][0] if context in self._args else (C.c_void_p * len(self.scenes))(*[s.handle for s in self.scenes])
        lib.check(self._l.rayca_hip_render_multi_wait(handles, len(self.scenes), context))


class SoftRenderer:
    """Drop-in for rayca_soft::SoftRenderer (scene.rs:11-14): `draw(scene, image)` renders `scene`
    with camera_draw_infos[0] into `image` (RGBA8).  Like the reference it rebuilds the BVH on
    every draw; use DeviceScene directly to amortise the build."""

    def __init__(self, config: Optional[Config] = None, device: int = 0):
        self.config = config or Config()
        self.device = device
        self.last_stats = None

    @staticmethod
    def new_with_config(config: Config) -> "SoftRenderer":
        return SoftRenderer(config)

    @staticmethod
    def create_default_model():
        return create_default_model()

    def draw(self, scene: Scene, image: Image) -> None:
        if image.color_type != abi.COLOR_RGBA8:
            raise ValueError("draw() writes RGBA8 images (scene.rs:117)")
        ds = DeviceScene(flatten(scene), self.config, self.device)
        try:
            u8, _, self.last_stats = ds.render(self.config, image.width, image.height, want_f32=False)
            image.data[...] = u8
        finally:
            ds.close()
