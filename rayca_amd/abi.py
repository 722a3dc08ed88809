"""ctypes mirror of include/rayca_hip.h (the C ABI of librayca_hip.so).

Pure data-layout code: no compute.  `SceneDesc` owns the numpy buffers a RaycaSceneDesc points to,
so the pointers stay valid for as long as the Python object lives.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

ABI_VERSION = 2
NONE = 0xFFFFFFFF

# status codes
OK = 0
ERR_BAD_ARG = -1
ERR_NO_CAMERA = -2
ERR_EMPTY_SCENE = -3
ERR_HIP = -4
ERR_OOM = -5
ERR_UNSUPPORTED = -6
ERR_NO_DEVICE = -7
ERR_BVH_DEPTH = -8
ERR_RCCL = -9

# rayca-soft/src/integrator/mod.rs:32-41
INTEGRATOR_SCRATCHER, INTEGRATOR_RAYTRACER, INTEGRATOR_FLAT = 0, 1, 2
INTEGRATOR_ANALYTIC_DIRECT, INTEGRATOR_DIRECT, INTEGRATOR_PATHTRACER = 3, 4, 5
# rayca-soft/src/sampler/mod.rs:41-50
SAMPLER_NONE, SAMPLER_NEE, SAMPLER_HEMISPHERE, SAMPLER_COSINE, SAMPLER_BRDF, SAMPLER_MIS = range(6)
MATERIAL_PBR, MATERIAL_PHONG, MATERIAL_GGX = 0, 1, 2
LIGHT_DIRECTIONAL, LIGHT_POINT, LIGHT_QUAD = 0, 1, 2
GEOMETRY_TRIANGLE_MESH, GEOMETRY_SPHERE = 0, 1
INDEX_U8, INDEX_U16, INDEX_U32 = 5121, 5123, 5125
COLOR_RGB8, COLOR_RGBA8, COLOR_RGBA32F = 0, 1, 2
BUILDER_REFERENCE, BUILDER_SAH = 0, 1
ENGINE_AUTO, ENGINE_GENERAL, ENGINE_WAVEFRONT, ENGINE_FUSED = 0, 1, 2, 3
TRAVERSAL_ORDERED, TRAVERSAL_EXHAUSTIVE = 0, 1
CAMERA_AUTO, CAMERA_GENERATION, CAMERA_REFILL = 0, 1, 2
# RaycaStats.class_ms / class_launches
KERNEL_GENERATION, KERNEL_FLAT_REFILL, KERNEL_WF_TRACE, KERNEL_QUEUE_REFILL, KERNEL_WF_SHADE, KERNEL_WF_SHADOW, KERNEL_SHADOW_REFILL, KERNEL_OTHER = range(8)
KERNEL_NAMES = ("k_generation", "k_flat_refill", "k_wf_trace", "k_queue_refill", "k_wf_shade", "k_wf_shadow", "k_shadow_refill", "other")
GATHER_RCCL, GATHER_PEER_COPY = 0, 1


class RaycaConfig(C.Structure):
    _fields_ = [
        ("bvh", C.c_uint32),
        ("light_samples", C.c_uint32),
        ("light_stratify", C.c_uint32),
        ("samples_per_pixel", C.c_uint32),
        ("russian_roulette", C.c_uint32),
        ("direct_sampler", C.c_uint32),
        ("indirect_sampler", C.c_uint32),
        ("integrator", C.c_uint32),
        ("max_depth", C.c_uint32),
        ("gamma", C.c_float),
        ("seed", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class RaycaTrs(C.Structure):
    _fields_ = [("translation", C.c_float * 3), ("rotation", C.c_float * 4), ("scale", C.c_float * 3)]


class RaycaNode(C.Structure):
    _fields_ = [
        ("parent", C.c_int32),
        ("model", C.c_uint32),
        ("mesh", C.c_uint32),
        ("camera", C.c_uint32),
        ("light", C.c_uint32),
        ("trs", RaycaTrs),
    ]


class RaycaMesh(C.Structure):
    _fields_ = [("first_primitive", C.c_uint32), ("primitive_count", C.c_uint32)]


class RaycaPrimitive(C.Structure):
    _fields_ = [
        ("geometry", C.c_uint32),
        ("material", C.c_uint32),
        ("first_vertex", C.c_uint32),
        ("vertex_count", C.c_uint32),
        ("index_byte_offset", C.c_uint64),
        ("index_count", C.c_uint32),
        ("index_type", C.c_uint32),
        ("sphere_center", C.c_float * 3),
        ("sphere_radius", C.c_float),
    ]


class RaycaMaterial(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32),
        ("albedo_texture", C.c_uint32),
        ("normal_texture", C.c_uint32),
        ("metallic_roughness_texture", C.c_uint32),
        ("color", C.c_float * 4),
        ("metallic_factor", C.c_float),
        ("roughness_factor", C.c_float),
        ("shininess", C.c_float),
        ("pad0", C.c_float),
        ("ambient", C.c_float * 4),
        ("emission", C.c_float * 4),
        ("diffuse", C.c_float * 4),
        ("specular", C.c_float * 4),
    ]


class RaycaTexture(C.Structure):
    _fields_ = [("image", C.c_uint32)]


class RaycaImage(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("color_type", C.c_uint32),
        ("pad0", C.c_uint32),
        ("byte_offset", C.c_uint64),
    ]


class RaycaCamera(C.Structure):
    _fields_ = [("yfov_radians", C.c_float)]


class RaycaLight(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32),
        ("material", C.c_uint32),
        ("intensity", C.c_float),
        ("pad0", C.c_float),
        ("color", C.c_float * 4),
        ("attenuation", C.c_float * 3),
        ("pad1", C.c_float),
        ("ab", C.c_float * 3),
        ("pad2", C.c_float),
        ("ac", C.c_float * 3),
        ("pad3", C.c_float),
    ]


class RaycaSceneDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("flags", C.c_uint32),
        ("nodes", C.POINTER(RaycaNode)),
        ("node_count", C.c_uint32),
        ("meshes", C.POINTER(RaycaMesh)),
        ("mesh_count", C.c_uint32),
        ("primitives", C.POINTER(RaycaPrimitive)),
        ("primitive_count", C.c_uint32),
        ("vertex_count", C.c_uint32),
        ("positions", C.POINTER(C.c_float)),
        ("colors", C.POINTER(C.c_float)),
        ("normals", C.POINTER(C.c_float)),
        ("tangents", C.POINTER(C.c_float)),
        ("bitangents", C.POINTER(C.c_float)),
        ("uvs", C.POINTER(C.c_float)),
        ("index_bytes", C.POINTER(C.c_uint8)),
        ("index_byte_count", C.c_uint64),
        ("materials", C.POINTER(RaycaMaterial)),
        ("material_count", C.c_uint32),
        ("textures", C.POINTER(RaycaTexture)),
        ("texture_count", C.c_uint32),
        ("images", C.POINTER(RaycaImage)),
        ("image_count", C.c_uint32),
        ("image_bytes", C.POINTER(C.c_uint8)),
        ("image_byte_count", C.c_uint64),
        ("cameras", C.POINTER(RaycaCamera)),
        ("camera_count", C.c_uint32),
        ("lights", C.POINTER(RaycaLight)),
        ("light_count", C.c_uint32),
    ]


class RaycaBuildOptions(C.Structure):
    _fields_ = [("builder", C.c_uint32), ("device", C.c_uint32), ("build_on_host", C.c_uint32), ("reserved", C.c_uint32 * 5)]


class RaycaTile(C.Structure):
    _fields_ = [("part", C.c_uint32), ("parts", C.c_uint32), ("band_rows", C.c_uint32), ("reserved", C.c_uint32)]


class RaycaRenderOptions(C.Structure):
    _fields_ = [
        ("traversal", C.c_uint32),
        ("collect_stats", C.c_uint32),
        ("tile", RaycaTile),
        ("stream", C.c_void_p),
        ("engine", C.c_uint32),
        ("context", C.c_uint32),
        ("camera_rays", C.c_uint32),
        ("reserved", C.c_uint32),
        ("wait_event", C.c_void_p),
        ("record_event", C.c_void_p),
    ]


class RaycaMultiOptions(C.Structure):
    _fields_ = [
        ("traversal", C.c_uint32),
        ("collect_stats", C.c_uint32),
        ("band_rows", C.c_uint32),
        ("gather", C.c_uint32),
        ("engine", C.c_uint32),
        ("output_on_device", C.c_uint32),
        ("context", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class RaycaStats(C.Structure):
    _fields_ = [
        ("rays_primary", C.c_uint64),
        ("rays_shadow", C.c_uint64),
        ("rays_bounce", C.c_uint64),
        ("boxes_tested", C.c_uint64),
        ("triangles_tested", C.c_uint64),
        ("hits_shaded", C.c_uint64),
        ("wave_box_slots", C.c_uint64),
        ("wave_triangle_slots", C.c_uint64),
        ("kernel_ms", C.c_float),
        ("trace_kernel_ms", C.c_float),
        ("kernel_launches", C.c_uint32),
        ("trace_kernel_launches", C.c_uint32),
        ("rows_rendered", C.c_uint32),
        ("node_format", C.c_uint32),
        ("class_ms", C.c_float * 8),
        ("class_launches", C.c_uint32 * 8),
    ]

    def as_dict(self):
        return {n: (list(getattr(self, n)) if n.startswith("class_") else getattr(self, n)) for n, _ in self._fields_}


class RaycaSceneInfo(C.Structure):
    _fields_ = [
        ("triangle_count", C.c_uint32),
        ("sphere_count", C.c_uint32),
        ("blas_count", C.c_uint32),
        ("node_count", C.c_uint32),
        ("max_depth", C.c_uint32),
        ("light_count", C.c_uint32),
        ("device_bytes", C.c_uint64),
        ("build_ms", C.c_float),
        ("runtime_init_ms", C.c_float),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None and a.size else None


def _u8ptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8)) if a is not None and a.size else None


def _array(ctype, items):
    arr = (ctype * max(len(items), 1))()
    for i, it in enumerate(items):
        arr[i] = it
    return arr


class SceneDesc:
    """Owns the buffers behind one RaycaSceneDesc."""

    def __init__(self, *, nodes, meshes, primitives, positions, colors=None, normals=None,
                 tangents=None, bitangents=None, uvs=None, index_bytes=None, materials=(),
                 textures=(), images=(), image_bytes=None, cameras=(), lights=()):
        def f32(a, width):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1, width)
            return a

        self.positions = f32(positions, 3)
        if self.positions is None:
            self.positions = np.zeros((0, 3), np.float32)
        n = self.positions.shape[0]
        self.colors = f32(colors, 4)
        self.normals = f32(normals, 3)
        self.tangents = f32(tangents, 3)
        self.bitangents = f32(bitangents, 3)
        self.uvs = f32(uvs, 2)
        for name in ("colors", "normals", "tangents", "bitangents", "uvs"):
            a = getattr(self, name)
            if a is not None and a.shape[0] != n:
                raise ValueError(f"{name}: expected {n} vertices, got {a.shape[0]}")
        self.index_bytes = np.ascontiguousarray(
            index_bytes if index_bytes is not None else np.zeros(0, np.uint8), dtype=np.uint8)
        self.image_bytes = np.ascontiguousarray(
            image_bytes if image_bytes is not None else np.zeros(0, np.uint8), dtype=np.uint8)
        self._nodes = _array(RaycaNode, list(nodes))
        self._meshes = _array(RaycaMesh, list(meshes))
        self._prims = _array(RaycaPrimitive, list(primitives))
        self._materials = _array(RaycaMaterial, list(materials))
        self._textures = _array(RaycaTexture, list(textures))
        self._images = _array(RaycaImage, list(images))
        self._cameras = _array(RaycaCamera, list(cameras))
        self._lights = _array(RaycaLight, list(lights))
        d = RaycaSceneDesc()
        d.abi_version = ABI_VERSION
        d.flags = 0
        d.nodes, d.node_count = self._nodes, len(nodes)
        d.meshes, d.mesh_count = self._meshes, len(meshes)
        d.primitives, d.primitive_count = self._prims, len(primitives)
        d.vertex_count = n
        d.positions = _fptr(self.positions)
        d.colors = _fptr(self.colors)
        d.normals = _fptr(self.normals)
        d.tangents = _fptr(self.tangents)
        d.bitangents = _fptr(self.bitangents)
        d.uvs = _fptr(self.uvs)
        d.index_bytes = _u8ptr(self.index_bytes)
        d.index_byte_count = self.index_bytes.size
        d.materials, d.material_count = self._materials, len(materials)
        d.textures, d.texture_count = self._textures, len(textures)
        d.images, d.image_count = self._images, len(images)
        d.image_bytes = _u8ptr(self.image_bytes)
        d.image_byte_count = self.image_bytes.size
        d.cameras, d.camera_count = self._cameras, len(cameras)
        d.lights, d.light_count = self._lights, len(lights)
        self.c = d

    def ptr(self):
        return C.byref(self.c)


def bind_product_signatures(lib):
    """argtypes/restype for every entry point declared in include/rayca_hip.h."""
    P = C.POINTER
    lib.rayca_hip_version.restype = C.c_uint32
    lib.rayca_hip_version.argtypes = []
    lib.rayca_hip_device_count.restype = C.c_int32
    lib.rayca_hip_device_count.argtypes = []
    lib.rayca_hip_selftest.restype = C.c_int32
    lib.rayca_hip_selftest.argtypes = []
    lib.rayca_hip_last_error.restype = None
    lib.rayca_hip_last_error.argtypes = [C.c_char_p, C.c_size_t]
    lib.rayca_hip_config_default.restype = None
    lib.rayca_hip_config_default.argtypes = [P(RaycaConfig)]
    lib.rayca_hip_scene_create.restype = C.c_int32
    lib.rayca_hip_scene_create.argtypes = [P(RaycaSceneDesc), P(RaycaConfig), P(RaycaBuildOptions), P(C.c_void_p)]
    lib.rayca_hip_scene_destroy.restype = C.c_int32
    lib.rayca_hip_scene_destroy.argtypes = [C.c_void_p]
    lib.rayca_hip_scene_reap.restype = C.c_int32
    lib.rayca_hip_scene_reap.argtypes = []
    lib.rayca_hip_scene_info.restype = C.c_int32
    lib.rayca_hip_scene_info.argtypes = [C.c_void_p, P(RaycaSceneInfo)]
    lib.rayca_hip_scene_finish.restype = C.c_int32
    lib.rayca_hip_scene_finish.argtypes = [C.c_void_p]
    lib.rayca_hip_render.restype = C.c_int32
    lib.rayca_hip_render.argtypes = [C.c_void_p, P(RaycaConfig), C.c_uint32, C.c_uint32, P(RaycaRenderOptions),
                                     C.c_void_p, C.c_void_p, P(RaycaStats)]
    lib.rayca_hip_render_device.restype = C.c_int32
    lib.rayca_hip_render_device.argtypes = [C.c_void_p, P(RaycaConfig), C.c_uint32, C.c_uint32,
                                            P(RaycaRenderOptions), C.c_void_p, C.c_void_p, P(RaycaStats)]
    lib.rayca_hip_tile_rows.restype = C.c_uint32
    lib.rayca_hip_tile_rows.argtypes = [P(RaycaTile), C.c_uint32]
    lib.rayca_hip_trace_rays.restype = C.c_int32
    lib.rayca_hip_trace_rays.argtypes = [C.c_void_p, P(RaycaRenderOptions), C.c_uint32, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, P(RaycaStats)]
    lib.rayca_hip_scene_primitive_order.restype = C.c_int32
    lib.rayca_hip_scene_primitive_order.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    lib.rayca_hip_scene_read_nodes.restype = C.c_int32
    lib.rayca_hip_scene_read_nodes.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.rayca_hip_rccl_status.restype = C.c_int32
    lib.rayca_hip_rccl_status.argtypes = []
    lib.rayca_hip_render_multi.restype = C.c_int32
    lib.rayca_hip_render_multi.argtypes = [P(C.c_void_p), C.c_uint32, P(RaycaConfig), C.c_uint32, C.c_uint32, P(RaycaMultiOptions),
                                           C.c_void_p, P(RaycaStats)]
    lib.rayca_hip_render_multi_issue.restype = C.c_int32
    lib.rayca_hip_render_multi_issue.argtypes = [P(C.c_void_p), C.c_uint32, P(RaycaConfig), C.c_uint32, C.c_uint32, P(RaycaMultiOptions), C.c_void_p]
    lib.rayca_hip_render_multi_wait.restype = C.c_int32
    lib.rayca_hip_render_multi_wait.argtypes = [P(C.c_void_p), C.c_uint32, C.c_uint32]
    return lib


PRODUCT_SYMBOLS = [
    "rayca_hip_version", "rayca_hip_device_count", "rayca_hip_selftest", "rayca_hip_last_error", "rayca_hip_config_default",
    "rayca_hip_scene_create", "rayca_hip_scene_destroy", "rayca_hip_scene_reap", "rayca_hip_scene_info", "rayca_hip_scene_finish", "rayca_hip_render",
    "rayca_hip_render_device", "rayca_hip_tile_rows", "rayca_hip_trace_rays",
    "rayca_hip_scene_primitive_order", "rayca_hip_scene_read_nodes", "rayca_hip_render_multi", "rayca_hip_render_multi_issue", "rayca_hip_render_multi_wait",
    "rayca_hip_rccl_status",
]
