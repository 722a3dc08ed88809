"""ctypes binding of oracle/_build/librayca_oracle.so -- TEST INFRASTRUCTURE.

The oracle is the CPU restatement of the reference's algorithm.  It may be imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg only; never by the product package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from rayca_amd import abi

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(_ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "librayca_oracle.so")

BUILD_LITERAL, BUILD_BINNED = 0, 1
XFORM_PER_TEST, XFORM_CACHED = 0, 1


class OracleOptions(C.Structure):
    _fields_ = [("build", C.c_uint32), ("xform", C.c_uint32), ("threads", C.c_uint32), ("reserved", C.c_uint32)]


class OracleBvhNode(C.Structure):
    _fields_ = [("a", C.c_float * 4), ("b", C.c_float * 4), ("offset", C.c_uint32), ("count", C.c_uint32)]


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    src = [os.path.join(ORACLE_DIR, f) for f in ("rayca_oracle.c", "rayca_oracle.h", "oracle_math.h")]
    if not os.path.exists(LIB_PATH) or any(os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src):
        build()
    lib = C.CDLL(LIB_PATH)
    P = C.POINTER
    lib.oracle_last_error.restype = C.c_char_p
    lib.oracle_scene_create.restype = C.c_int32
    lib.oracle_scene_create.argtypes = [P(abi.RaycaSceneDesc), P(abi.RaycaConfig), P(OracleOptions), P(C.c_void_p)]
    lib.oracle_scene_destroy.argtypes = [C.c_void_p]
    lib.oracle_scene_destroy.restype = None
    for name in ("oracle_scene_blas_count", "oracle_scene_primitive_count"):
        getattr(lib, name).restype = C.c_uint32
        getattr(lib, name).argtypes = [C.c_void_p]
    for name in ("oracle_blas_node_count", "oracle_blas_primitive_count"):
        getattr(lib, name).restype = C.c_uint32
        getattr(lib, name).argtypes = [C.c_void_p, C.c_uint32]
    lib.oracle_blas_nodes.restype = C.c_int32
    lib.oracle_blas_nodes.argtypes = [C.c_void_p, C.c_uint32, P(OracleBvhNode), C.c_uint32]
    lib.oracle_scene_primitive_order.restype = C.c_int32
    lib.oracle_scene_primitive_order.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    lib.oracle_scene_world_triangles.restype = C.c_int32
    lib.oracle_scene_world_triangles.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    lib.oracle_render.restype = C.c_int32
    lib.oracle_render.argtypes = [C.c_void_p, P(abi.RaycaConfig), C.c_uint32, C.c_uint32, P(abi.RaycaTile),
                                  C.c_void_p, C.c_void_p, P(abi.RaycaStats), P(C.c_double)]
    lib.oracle_render_rows.restype = C.c_int32
    lib.oracle_render_rows.argtypes = [C.c_void_p, P(abi.RaycaConfig), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                       C.c_void_p, C.c_void_p, P(abi.RaycaStats), P(C.c_double)]
    lib.oracle_trace_rays.restype = C.c_int32
    lib.oracle_trace_rays.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      P(abi.RaycaStats)]
    F3, F4 = C.c_float * 3, C.c_float * 4
    lib.oracle_vec3_dot.restype = C.c_float
    lib.oracle_aabb_intersects.restype = C.c_float
    lib.oracle_sampler_wrap.restype = C.c_uint32
    lib.oracle_sampler_wrap.argtypes = [C.c_float, C.c_uint32]
    lib.oracle_rng_root.restype = C.c_uint32
    lib.oracle_rng_root.argtypes = [C.c_uint32] * 3
    lib.oracle_rng_child.restype = C.c_uint32
    lib.oracle_rng_child.argtypes = [C.c_uint32] * 2
    lib.oracle_rng_f32.restype = C.c_float
    lib.oracle_rng_f32.argtypes = [C.c_uint32] * 2
    lib.oracle_sphere_intersects.argtypes = [C.c_void_p, C.c_float, P(abi.RaycaTrs), C.c_void_p, C.c_void_p,
                                             P(C.c_float), C.c_void_p]
    VP = C.c_void_p
    lib.oracle_sphere_bounds.argtypes = [VP, C.c_float, P(abi.RaycaTrs), VP, VP, VP]
    lib.oracle_triangle_bounds.argtypes = [VP, P(abi.RaycaTrs), VP, VP, VP]
    lib.oracle_scene_primitive_intersects.restype = C.c_int32
    lib.oracle_scene_primitive_intersects.argtypes = [VP, C.c_uint32, VP, VP, P(C.c_float), VP]
    lib.oracle_quat_axis_angle.argtypes = [VP, C.c_float, VP]
    lib.oracle_quat_dot.restype = C.c_float
    lib.oracle_quat_len.restype = C.c_float
    lib.oracle_vec3_arith.argtypes = [VP, VP, C.c_float] + [VP] * 5
    _lib = lib
    return lib


class OracleError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"oracle error {code}: {msg}")
        self.code = code


def _check(rc):
    if rc != 0:
        raise OracleError(rc, load().oracle_last_error().decode())


def f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def f4(v):
    return (C.c_float * 4)(*[float(x) for x in v])


def trs(translation=(0, 0, 0), rotation=(0, 0, 0, 1), scale=(1, 1, 1)):
    t = abi.RaycaTrs()
    t.translation[:], t.rotation[:], t.scale[:] = translation, rotation, scale
    return t


class OracleScene:
    def __init__(self, desc: abi.SceneDesc, config=None, build=BUILD_LITERAL, xform=XFORM_CACHED, threads=0):
        self.lib = load()
        self.desc = desc
        cfg = config.to_abi() if config is not None else None
        o = OracleOptions(build, xform, threads, 0)
        h = C.c_void_p()
        _check(self.lib.oracle_scene_create(desc.ptr(), C.byref(cfg) if cfg is not None else None, C.byref(o), C.byref(h)))
        self.handle = h
        self.threads = threads

    def close(self):
        if getattr(self, "handle", None):
            self.lib.oracle_scene_destroy(self.handle)
            self.handle = None

    __del__ = close

    @property
    def blas_count(self):
        return self.lib.oracle_scene_blas_count(self.handle)

    @property
    def primitive_count(self):
        return self.lib.oracle_scene_primitive_count(self.handle)

    def blas_nodes(self, blas):
        n = self.lib.oracle_blas_node_count(self.handle, blas)
        arr = (OracleBvhNode * max(n, 1))()
        _check(self.lib.oracle_blas_nodes(self.handle, blas, arr, n))
        a = np.array([[*x.a, *x.b] for x in arr[:n]], np.float32).reshape(n, 8)
        rng = np.array([[x.offset, x.count] for x in arr[:n]], np.uint32).reshape(n, 2)
        return a, rng

    def primitive_order(self):
        n = self.primitive_count
        out = np.zeros(n, np.uint32)
        _check(self.lib.oracle_scene_primitive_order(self.handle, out.ctypes.data, n))
        return out

    def world_triangles(self, flat_count):
        out = np.zeros((flat_count, 9), np.float32)
        _check(self.lib.oracle_scene_world_triangles(self.handle, out.ctypes.data, flat_count))
        return out

    def render(self, config, width, height, tile=None, rows=None, want_rgba8=True, want_f32=True):
        cfg = config.to_abi()
        st = abi.RaycaStats()
        secs = C.c_double()
        if rows is not None:
            nrows = rows[1] - rows[0]
        elif tile is not None:
            band = tile[2] or 1
            nrows = sum(1 for y in range(height) if (y // band) % tile[1] == tile[0]) if tile[1] > 1 else height
        else:
            nrows = height
        u8 = np.zeros((nrows, width, 4), np.uint8) if want_rgba8 else None
        f32 = np.zeros((nrows, width, 4), np.float32) if want_f32 else None
        pu8 = u8.ctypes.data if u8 is not None else None
        pf = f32.ctypes.data if f32 is not None else None
        if rows is not None:
            _check(self.lib.oracle_render_rows(self.handle, C.byref(cfg), width, height, rows[0], rows[1], pu8, pf,
                                               C.byref(st), C.byref(secs)))
        else:
            t = None
            if tile is not None:
                t = abi.RaycaTile()
                t.part, t.parts, t.band_rows = tile
            _check(self.lib.oracle_render(self.handle, C.byref(cfg), width, height, C.byref(t) if t is not None else None,
                                          pu8, pf, C.byref(st), C.byref(secs)))
        d = st.as_dict()
        d["seconds"] = secs.value
        return u8, f32, d

    def trace_rays(self, rays):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        n = rays.shape[0]
        t = np.zeros(n, np.float32)
        prim = np.zeros(n, np.uint32)
        uv = np.zeros((n, 2), np.float32)
        st = abi.RaycaStats()
        _check(self.lib.oracle_trace_rays(self.handle, n, rays.ctypes.data, t.ctypes.data, prim.ctypes.data,
                                          uv.ctypes.data, C.byref(st)))
        return t, prim, uv, st.as_dict()
