"""Host-side mirror of the reference's scene types for the hot path.

Same names and argument meaning as rayca-model / rayca-soft so that tests read like the reference's
own tests (rayca-soft/tests/gltf.rs): Scene, Model, Node, Mesh, Primitive, TriangleMesh, Sphere,
Camera, Light, PbrMaterial/PhongMaterial/GgxMaterial, Trs, Image.  Handles (rayca-util Pack/Handle)
are plain list indices.  Nothing here computes: `flatten()` only serialises the graph into the flat
RaycaSceneDesc that crosses the C ABI; world transforms, BVH and shading all happen behind it.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import abi


# ---- rayca-math ---------------------------------------------------------------------------------
@dataclass
class Trs:
    """rayca_math::Trs (rayca-math/src/trs.rs:75-86): scale -> rotate -> translate."""
    translation: tuple = (0.0, 0.0, 0.0)
    rotation: tuple = (0.0, 0.0, 0.0, 1.0)  # quaternion x, y, z, w
    scale: tuple = (1.0, 1.0, 1.0)

    def to_abi(self) -> abi.RaycaTrs:
        t = abi.RaycaTrs()
        t.translation[:] = [float(np.float32(v)) for v in self.translation]
        t.rotation[:] = [float(np.float32(v)) for v in self.rotation]
        t.scale[:] = [float(np.float32(v)) for v in self.scale]
        return t


def quat_axis_angle(axis, angle_radians):
    """Quat::axis_angle (rayca-math/src/quat.rs:66-75), evaluated in f32."""
    f32 = np.float32
    half = f32(angle_radians) / f32(2.0)
    s, c = f32(math.sin(float(half))), f32(math.cos(float(half)))
    q = np.array([f32(axis[0]) * s, f32(axis[1]) * s, f32(axis[2]) * s, c], dtype=np.float32)
    n = f32(math.sqrt(float((q * q).sum(dtype=np.float32))))
    q = q / n
    return tuple(float(v) for v in q)


# ---- rayca-geometry -----------------------------------------------------------------------------
@dataclass
class TriangleMesh:
    """rayca_geometry::TriangleMesh (triangle.rs:309-314) with SoA numpy vertex attributes.
    Missing attributes take Vertex::default() (vertex.rs:164-175)."""
    positions: np.ndarray  # (N,3) f32
    indices: np.ndarray    # (M,) u8/u16/u32
    colors: Optional[np.ndarray] = None
    normals: Optional[np.ndarray] = None
    tangents: Optional[np.ndarray] = None
    bitangents: Optional[np.ndarray] = None
    uvs: Optional[np.ndarray] = None

    @staticmethod
    def unit():
        """TriangleMesh::unit (triangle.rs:327-342)."""
        return TriangleMesh(np.array([[-1, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32),
                            np.array([0, 1, 2], np.uint8))

    @staticmethod
    def quad(uv_scale=(1.0, 1.0)):
        """TriangleMesh::quad (triangle.rs:344-378)."""
        pos = np.array([[-0.5, -0.5, 0], [0.5, -0.5, 0], [0.5, 0.5, 0], [-0.5, 0.5, 0]], np.float32)
        uv = np.array([[0, 1], [1, 1], [1, 0], [0, 0]], np.float32) * np.array(uv_scale, np.float32)
        nrm = np.tile(np.array([[0, 0, 1]], np.float32), (4, 1))
        return TriangleMesh(pos, np.array([0, 1, 2, 2, 3, 0], np.uint8), normals=nrm, uvs=uv,
                            colors=np.ones((4, 4), np.float32))

    @staticmethod
    def cube():
        """TriangleMesh::cube (triangle.rs:380-548)."""
        faces = [
            ((0, 0, 1), [(-.5, -.5, .5), (.5, -.5, .5), (.5, .5, .5), (-.5, .5, .5)]),
            ((1, 0, 0), [(.5, -.5, .5), (.5, -.5, -.5), (.5, .5, -.5), (.5, .5, .5)]),
            ((0, 0, -1), [(.5, -.5, -.5), (-.5, -.5, -.5), (-.5, .5, -.5), (.5, .5, -.5)]),
            ((-1, 0, 0), [(-.5, -.5, -.5), (-.5, -.5, .5), (-.5, .5, .5), (-.5, .5, -.5)]),
            ((0, 1, 0), [(-.5, .5, .5), (.5, .5, .5), (.5, .5, -.5), (-.5, .5, -.5)]),
            ((0, -1, 0), [(-.5, -.5, -.5), (.5, -.5, -.5), (.5, -.5, .5), (-.5, -.5, .5)]),
        ]
        pos, nrm, uv, idx = [], [], [], []
        for f, (n, quad) in enumerate(faces):
            pos += quad
            nrm += [n] * 4
            uv += [(0, 0), (1, 0), (1, 1), (0, 1)]
            b = 4 * f
            idx += [b, b + 1, b + 2, b, b + 2, b + 3]
        return TriangleMesh(np.array(pos, np.float32), np.array(idx, np.uint8),
                            normals=np.array(nrm, np.float32), uvs=np.array(uv, np.float32),
                            colors=np.ones((24, 4), np.float32))


@dataclass
class Sphere:
    """rayca_geometry::Sphere (sphere.rs:38-44)."""
    center: tuple = (0.0, 0.0, 0.0)
    radius: float = 1.0

    @staticmethod
    def unit():
        return Sphere()


# ---- rayca-model --------------------------------------------------------------------------------
@dataclass
class PbrMaterial:
    """material/pbr.rs:58-66."""
    color: tuple = (1.0, 1.0, 1.0, 1.0)
    albedo: Optional[int] = None
    normal: Optional[int] = None
    metallic_factor: float = 0.0
    roughness_factor: float = 0.0
    metallic_roughness: Optional[int] = None


@dataclass
class PhongMaterial:
    """material/phong.rs:10-34."""
    ambient: tuple = (0.0, 0.0, 0.0, 1.0)
    emission: tuple = (0.0, 0.0, 0.0, 1.0)
    diffuse: tuple = (0.0, 0.0, 0.0, 1.0)
    specular: tuple = (0.0, 0.0, 0.0, 1.0)
    shininess: float = 0.0


@dataclass
class GgxMaterial:
    """material/ggx.rs:10-23."""
    diffuse: tuple = (0.0, 0.0, 0.0, 1.0)
    specular: tuple = (0.0, 0.0, 0.0, 1.0)
    roughness: float = 0.0


@dataclass
class Texture:
    image: int = 0


@dataclass
class Image:
    """rayca_model::Image (image.rs:26-36): row-major, top-left origin."""
    width: int
    height: int
    color_type: int = abi.COLOR_RGBA8
    data: Optional[np.ndarray] = None

    def __post_init__(self):
        ch = {abi.COLOR_RGB8: 3, abi.COLOR_RGBA8: 4, abi.COLOR_RGBA32F: 4}[self.color_type]
        dt = np.float32 if self.color_type == abi.COLOR_RGBA32F else np.uint8
        if self.data is None:
            self.data = np.zeros((self.height, self.width, ch), dt)
        else:
            self.data = np.ascontiguousarray(self.data, dt).reshape(self.height, self.width, ch)


@dataclass
class Camera:
    """camera.rs:20-24; Camera::default() = infinite_perspective(1, pi/4, 0.1)."""
    yfov_radians: float = math.pi / 4


@dataclass
class Light:
    """light/mod.rs:15-19 flattened: kind + the union of the three payloads."""
    kind: int = abi.LIGHT_DIRECTIONAL
    color: tuple = (1.0, 1.0, 1.0, 1.0)
    intensity: float = 1.0
    attenuation: tuple = (0.0, 0.0, 1.0)  # point.rs:20
    ab: tuple = (0.0, 0.0, 0.0)
    ac: tuple = (0.0, 0.0, 0.0)
    material: Optional[int] = None

    @staticmethod
    def point():
        return Light(kind=abi.LIGHT_POINT)

    @staticmethod
    def quad(ab, ac, color=(1.0, 1.0, 1.0, 1.0), material=None, intensity=1.0):
        return Light(kind=abi.LIGHT_QUAD, ab=tuple(ab), ac=tuple(ac), color=tuple(color),
                     material=material, intensity=intensity)

    def set_intensity(self, v):
        self.intensity = v


@dataclass
class Primitive:
    geometry: int
    material: Optional[int] = None


@dataclass
class Mesh:
    primitives: List[int] = field(default_factory=list)


@dataclass
class Node:
    """node.rs:11-32."""
    trs: Trs = field(default_factory=Trs)
    children: List[int] = field(default_factory=list)
    mesh: Optional[int] = None
    camera: Optional[int] = None
    light: Optional[int] = None
    model: Optional[int] = None
    name: Optional[str] = None


class _Pack(list):
    def push(self, item) -> int:
        self.append(item)
        return len(self) - 1


class Model:
    """model.rs:30-50."""

    def __init__(self, name="Unknown"):
        self.name = name
        self.root = Node()
        self.nodes = _Pack()
        self.meshes = _Pack()
        self.primitives = _Pack()
        self.geometries = _Pack()
        self.materials = _Pack()  # PbrMaterial | PhongMaterial | GgxMaterial
        self.textures = _Pack()
        self.images = _Pack()
        self.cameras = _Pack()
        self.lights = _Pack()


class Scene:
    """scene.rs:47-53."""

    def __init__(self, name="Unknown"):
        self.name = name
        self.nodes = _Pack()
        self.models = _Pack()
        self.root = Node()

    def push_model(self, model: Model) -> int:
        """Scene::push_model (scene.rs:107-113)."""
        h = self.models.push(model)
        n = self.nodes.push(Node(model=h))
        self.root.children.append(n)
        return n


def create_default_model() -> Model:
    """SoftRenderer::create_default_model (rayca-soft/src/scene.rs:18-55)."""
    model = Model()
    cam = model.cameras.push(Camera())
    n = model.nodes.push(Node(camera=cam, trs=Trs(translation=(0.0, 0.0, 4.0))))
    model.root.children.append(n)
    light = Light.point()
    light.set_intensity(1024.0)
    lh = model.lights.push(light)
    n = model.nodes.push(Node(light=lh, trs=Trs(translation=(-1.0, 4.0, 3.0))))
    model.root.children.append(n)
    n = model.nodes.push(Node(light=lh, trs=Trs(translation=(1.0, 4.0, 3.0))))
    model.root.children.append(n)
    return model


# ---- flatten: Scene -> RaycaSceneDesc -----------------------------------------------------------
def _opt(h):
    return abi.NONE if h is None else int(h)


def _material_to_abi(m, tex_base):
    r = abi.RaycaMaterial()
    r.albedo_texture = r.normal_texture = r.metallic_roughness_texture = abi.NONE
    r.color[:] = (1, 1, 1, 1)
    for name in ("ambient", "emission", "diffuse", "specular"):
        getattr(r, name)[:] = (0, 0, 0, 1)
    r.roughness_factor = 1.0

    def tex(h):
        return abi.NONE if h is None else tex_base + int(h)

    if isinstance(m, PbrMaterial):
        r.kind = abi.MATERIAL_PBR
        r.color[:] = m.color
        r.albedo_texture, r.normal_texture = tex(m.albedo), tex(m.normal)
        r.metallic_roughness_texture = tex(m.metallic_roughness)
        r.metallic_factor, r.roughness_factor = m.metallic_factor, m.roughness_factor
    elif isinstance(m, PhongMaterial):
        r.kind = abi.MATERIAL_PHONG
        r.ambient[:], r.emission[:] = m.ambient, m.emission
        r.diffuse[:], r.specular[:] = m.diffuse, m.specular
        r.shininess = m.shininess
    elif isinstance(m, GgxMaterial):
        r.kind = abi.MATERIAL_GGX
        r.diffuse[:], r.specular[:] = m.diffuse, m.specular
        r.roughness_factor = m.roughness
    else:
        raise TypeError(f"unknown material {type(m)}")
    return r


def flatten(scene: Scene) -> abi.SceneDesc:
    """Serialise the graph in the traversal order of SceneDrawInfo::traverse_scene
    (rayca-soft/src/scene.rs:206-282): DFS pre-order, a node's model before its children."""
    nodes, meshes, prims = [], [], []
    materials, textures, images, cameras, lights = [], [], [], [], []
    pos, col, nrm, tan, bit, uvs = [], [], [], [], [], []
    any_col = any_nrm = any_tan = any_bit = any_uv = False
    index_chunks, image_chunks = [], []
    index_off = image_off = vertex_off = 0
    model_base = {}

    def add_model_payload(mh):
        nonlocal index_off, image_off, vertex_off, any_col, any_nrm, any_tan, any_bit, any_uv
        if mh in model_base:
            return model_base[mh]
        model: Model = scene.models[mh]
        base = dict(mesh=len(meshes), material=len(materials), texture=len(textures), image=len(images),
                    camera=len(cameras), light=len(lights))
        for im in model.images:
            r = abi.RaycaImage()
            r.width, r.height, r.color_type, r.byte_offset = im.width, im.height, im.color_type, image_off
            raw = np.ascontiguousarray(im.data).view(np.uint8).reshape(-1)
            image_chunks.append(raw)
            image_off += raw.size
            images.append(r)
        for t in model.textures:
            r = abi.RaycaTexture()
            r.image = base["image"] + t.image
            textures.append(r)
        for m in model.materials:
            materials.append(_material_to_abi(m, base["texture"]))
        for c in model.cameras:
            r = abi.RaycaCamera()
            r.yfov_radians = c.yfov_radians
            cameras.append(r)
        for lt in model.lights:
            r = abi.RaycaLight()
            r.kind, r.intensity = lt.kind, lt.intensity
            r.material = abi.NONE if lt.material is None else base["material"] + lt.material
            r.color[:], r.attenuation[:] = lt.color, lt.attenuation
            r.ab[:], r.ac[:] = lt.ab, lt.ac
            lights.append(r)
        # geometries are emitted per primitive (a Primitive owns exactly one Geometry handle)
        for mesh in model.meshes:
            rm = abi.RaycaMesh()
            rm.first_primitive, rm.primitive_count = len(prims), len(mesh.primitives)
            for ph in mesh.primitives:
                p: Primitive = model.primitives[ph]
                g = model.geometries[p.geometry]
                rp = abi.RaycaPrimitive()
                rp.material = abi.NONE if p.material is None else base["material"] + p.material
                if isinstance(g, Sphere):
                    rp.geometry = abi.GEOMETRY_SPHERE
                    rp.sphere_center[:] = g.center
                    rp.sphere_radius = g.radius
                    rp.index_type = abi.INDEX_U32
                else:
                    rp.geometry = abi.GEOMETRY_TRIANGLE_MESH
                    n = int(np.asarray(g.positions).reshape(-1, 3).shape[0])
                    rp.first_vertex, rp.vertex_count = vertex_off, n
                    idx = np.ascontiguousarray(g.indices)
                    rp.index_type = {1: abi.INDEX_U8, 2: abi.INDEX_U16, 4: abi.INDEX_U32}[idx.dtype.itemsize]
                    raw = idx.view(np.uint8).reshape(-1)
                    pad = (-index_off) % 4
                    if pad:
                        index_chunks.append(np.zeros(pad, np.uint8))
                        index_off += pad
                    rp.index_byte_offset, rp.index_count = index_off, idx.size
                    index_chunks.append(raw)
                    index_off += raw.size
                    pos.append(np.asarray(g.positions, np.float32).reshape(-1, 3))
                    col.append(None if g.colors is None else np.asarray(g.colors, np.float32).reshape(-1, 4))
                    nrm.append(None if g.normals is None else np.asarray(g.normals, np.float32).reshape(-1, 3))
                    tan.append(None if g.tangents is None else np.asarray(g.tangents, np.float32).reshape(-1, 3))
                    bit.append(None if g.bitangents is None else np.asarray(g.bitangents, np.float32).reshape(-1, 3))
                    uvs.append(None if g.uvs is None else np.asarray(g.uvs, np.float32).reshape(-1, 2))
                    any_col |= g.colors is not None
                    any_nrm |= g.normals is not None
                    any_tan |= g.tangents is not None
                    any_bit |= g.bitangents is not None
                    any_uv |= g.uvs is not None
                    vertex_off += n
                prims.append(rp)
            meshes.append(rm)
        model_base[mh] = base
        return base

    def emit(node: Node, parent: int, model_id: int, base) -> int:
        r = abi.RaycaNode()
        r.parent, r.model, r.trs = parent, model_id, node.trs.to_abi()
        r.mesh = abi.NONE if node.mesh is None or base is None else base["mesh"] + node.mesh
        r.camera = abi.NONE if node.camera is None or base is None else base["camera"] + node.camera
        r.light = abi.NONE if node.light is None or base is None else base["light"] + node.light
        nodes.append(r)
        return len(nodes) - 1

    def walk_model_node(model: Model, mh: int, nh: int, parent: int, base):
        node = model.nodes[nh]
        me = emit(node, parent, mh, base)
        for c in node.children:
            walk_model_node(model, mh, c, me, base)

    def walk_scene_node(nh: int, parent: int):
        node = scene.nodes[nh]
        me = emit(Node(trs=node.trs), parent, abi.NONE, None)
        if node.model is not None:
            model = scene.models[node.model]
            base = add_model_payload(node.model)
            mroot = emit(Node(trs=model.root.trs), me, node.model, None)
            for c in model.root.children:
                walk_model_node(model, node.model, c, mroot, base)
        for c in node.children:
            walk_scene_node(c, me)

    root = emit(Node(trs=scene.root.trs), -1, abi.NONE, None)
    for c in scene.root.children:
        walk_scene_node(c, root)

    def cat(chunks, width, default, used):
        if not used or not chunks:
            return None
        out = []
        for p, c in zip(pos, chunks):
            out.append(np.tile(np.array([default], np.float32), (p.shape[0], 1)) if c is None else c)
        return np.concatenate(out, axis=0) if out else None

    positions = np.concatenate(pos, axis=0) if pos else np.zeros((0, 3), np.float32)
    return abi.SceneDesc(
        nodes=nodes, meshes=meshes, primitives=prims, positions=positions,
        colors=cat(col, 4, (1, 1, 1, 1), any_col), normals=cat(nrm, 3, (0, 0, 1), any_nrm),
        tangents=cat(tan, 3, (0, 0, 0), any_tan), bitangents=cat(bit, 3, (0, 0, 0), any_bit),
        uvs=cat(uvs, 2, (0, 0), any_uv),
        index_bytes=np.concatenate(index_chunks) if index_chunks else None,
        materials=materials, textures=textures, images=images,
        image_bytes=np.concatenate(image_chunks) if image_chunks else None,
        cameras=cameras, lights=lights)
