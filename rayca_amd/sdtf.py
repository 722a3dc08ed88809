"""SDTF ("UCSD scene description") loader and Config.apply: the Python mirror of include/rayca_sdtf.hpp.

reference: rayca-model/src/loader/sdtf.rs (SdtfConfig :89-122, parse_* :160-772, parse_line :774-830,
process_material / process_primitive :833-870, build :872-899), rayca-model/src/scene.rs:126-136
(push_sdtf_from_path), rayca-soft/src/config.rs:58-71 (Config::apply: maxdepth -1 -> 16).

Host side, before the C ABI.  The arithmetic that ends up in node transforms (Trs::left_mul trs.rs:111-118,
Quat::axis_angle quat.rs:67-77, Mat4::look_at -> Quat::from(&Mat4) -> get_inverse, Quat::angle_between quat.rs:118-127,
the face normal of `tri`) is evaluated with numpy float32 scalars in the reference's operation order; literals go
through libc's strtof and sin/cos through libm's sinf/cosf, like the C++ loader, so that both flatten a file to the same
bytes (tests/test_sdtf.py).  Quirks kept: see the header of include/rayca_sdtf.hpp.
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
from dataclasses import dataclass

import numpy as np

from . import abi
from .model import (Camera, GgxMaterial, Light, Mesh, Model, Node, PhongMaterial, Primitive, Scene, Sphere, TriangleMesh, Trs)

f32 = np.float32
_libm = C.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libc = C.CDLL(ctypes.util.find_library("c") or "libc.so.6")
for _n in ("sinf", "cosf"):
    getattr(_libm, _n).restype = C.c_float
    getattr(_libm, _n).argtypes = [C.c_float]
_libc.strtof.restype = C.c_float
_libc.strtof.argtypes = [C.c_char_p, C.POINTER(C.c_char_p)]

PI32 = f32(3.14159265358979323846)
EPS = f32(9.765625e-4)   # rayca-math/src/lib.rs:33


class SdtfError(ValueError):
    pass


class SdtfIntegratorStrategy:  # sdtf.rs:15-29
    Raytracer, AnalyticDirect, Direct, Pathtracer = range(4)


class SdtfSamplerStrategy:     # sdtf.rs:46-54
    NONE, Nee, Hemisphere, Cosine, Brdf, Mis = range(6)


class SdtfBrdfStrategy:        # sdtf.rs:71-75
    Phong, Ggx = range(2)


@dataclass
class SdtfConfig:              # sdtf.rs:89-122
    width: int = 0
    height: int = 0
    max_depth: int = 5
    light_samples: int = 1
    light_stratify: bool = False
    samples_per_pixel: int = 1
    direct_sampler: int = SdtfSamplerStrategy.NONE
    russian_roulette: bool = False
    indirect_sampler: int = SdtfSamplerStrategy.Hemisphere
    integrator: int = SdtfIntegratorStrategy.Raytracer
    brdf: int = SdtfBrdfStrategy.Phong
    gamma: float = 1.0


def apply(config, s: SdtfConfig):
    """Config::apply (rayca-soft/src/config.rs:58-71), in place; returns `config`.  bvh and russian_roulette are not
    taken from the file (the reference does not copy them either)."""
    from .renderer import IntegratorStrategy as I, SamplerStrategy as S
    config.max_depth = 16 if s.max_depth == -1 else int(np.uint32(s.max_depth))
    config.light_samples = s.light_samples
    config.light_stratify = s.light_stratify
    config.samples_per_pixel = s.samples_per_pixel
    to_sampler = {SdtfSamplerStrategy.NONE: S.NONE, SdtfSamplerStrategy.Nee: S.Nee, SdtfSamplerStrategy.Hemisphere: S.Hemisphere,
                  SdtfSamplerStrategy.Cosine: S.Cosine, SdtfSamplerStrategy.Brdf: S.Brdf, SdtfSamplerStrategy.Mis: S.Mis}
    config.direct_sampler = to_sampler[s.direct_sampler]
    config.indirect_sampler = to_sampler[s.indirect_sampler]
    config.integrator = {SdtfIntegratorStrategy.Raytracer: I.Raytracer, SdtfIntegratorStrategy.AnalyticDirect: I.AnalyticDirect,
                         SdtfIntegratorStrategy.Direct: I.Direct, SdtfIntegratorStrategy.Pathtracer: I.Pathtracer}[s.integrator]
    config.gamma = float(s.gamma)
    return config


# ---- rayca-math in float32 scalars, 4 lanes, the reference's operation order -----------------------------------------
def _v(x, y, z, w=0.0):
    return (f32(x), f32(y), f32(z), f32(w))


def _sum4(a):     # f32x4::reduce_sum: ordered
    return (((f32(-0.0) + a[0]) + a[1]) + a[2]) + a[3]


def _mul(a, b):
    return tuple(x * y for x, y in zip(a, b))


def _add(a, b):
    return tuple(x + y for x, y in zip(a, b))


def _sub(a, b):
    return tuple(x - y for x, y in zip(a, b))


def _scale(f, a):
    return tuple(f * x for x in a)


def _dot(a, b):
    return _sum4(_mul(a, b))


def _cross(a, b):  # vec3.rs:134-142
    t0 = (a[1], a[2], a[0], a[3])
    t1 = (b[2], b[0], b[1], b[3])
    t2, t3 = _mul(t0, b), _mul(t0, t1)
    return _sub(t3, (t2[1], t2[2], t2[0], t2[3]))


def _normalized(a):  # vec3.rs:183-193
    ln = np.sqrt(_dot(a, a))
    if ln > EPS:
        return (a[0] / ln, a[1] / ln, a[2] / ln, a[3] / f32(1.0))
    return a


def _rotate(v, q):  # vec3.rs:148-159
    u = (q[0] * f32(1.0), q[1] * f32(1.0), q[2] * f32(1.0), q[3] * f32(0.0))
    s = q[3]
    a = _scale(f32(2.0) * _dot(u, v), u)
    b = _scale(s * s - _dot(u, u), v)
    c = _scale(f32(2.0) * s, _cross(u, v))
    return _add(_add(a, b), c)


def _qmul(a, b):   # quat.rs:236-258
    return (a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0],
            -a[0] * b[2] + a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
            a[0] * b[1] - a[1] * b[0] + a[2] * b[3] + a[3] * b[2],
            -a[0] * b[0] - a[1] * b[1] - a[2] * b[2] + a[3] * b[3])


def _qlen(q):
    return np.sqrt(_sum4(_mul(q, q)))


def _qnormalized(q):
    ln = _qlen(q)
    return tuple(x / ln for x in q)


def _qinverse(q):  # quat.rs:99-103
    if not abs(_qlen(q) - f32(1.0)) < f32(0.001):
        raise SdtfError("inverse of a quaternion that is not normalised (quat.rs:101)")
    return (q[0] * f32(-1.0), q[1] * f32(-1.0), q[2] * f32(-1.0), q[3] * f32(1.0))


def _sinf(x):
    return f32(_libm.sinf(float(x)))


def _cosf(x):
    return f32(_libm.cosf(float(x)))


def _axis_angle(axis, angle):  # quat.rs:67-77
    factor = _sinf(angle / f32(2.0))
    a = _v(*axis)
    q = (a[0] * factor + f32(0.0), a[1] * factor + f32(0.0), a[2] * factor + f32(0.0), a[3] * factor + _cosf(angle / f32(2.0)))
    return _qnormalized(q)


def _angle_between(a, b):      # quat.rs:118-127
    a, b = _v(*a), _v(*b)
    c = _cross(a, b)
    w = np.sqrt(_dot(a, a) * _dot(b, b)) + _dot(a, b)
    return _qnormalized((c[0], c[1], c[2], w))


def _q(t: Trs):
    return tuple(f32(x) for x in t.rotation)


def _left_mul(self: Trs, rhs: Trs) -> Trs:  # trs.rs:111-118 (returns the new value of `self`)
    st, ss, sq = _v(*self.translation), _v(*self.scale), _q(self)
    rt, rs, rq = _v(*rhs.translation), _v(*rhs.scale), _q(rhs)
    t = _add(st, _rotate(_mul(ss, rt), sq))
    r = _qmul(sq, rq)
    s = _rotate(_mul(ss, _rotate(rs, rq)), _qinverse(rq))
    return Trs(translation=tuple(float(x) for x in t[:3]), rotation=tuple(float(x) for x in r), scale=tuple(float(x) for x in s[:3]))


def _look_at_rotation(target, eye, up):  # mat4.rs:81-95,117-119 ; quat.rs:184-226
    z = _normalized(_sub(_v(*eye), _v(*target)))
    x = _normalized(_cross(_v(*up), z))
    y = _cross(z, x)
    m = (x, y, z)
    one, two = f32(1.0), f32(2.0)
    t = m[0][0] + m[1][1] + m[2][2]
    if t > 0.0:
        s = f32(0.5) / np.sqrt(t + one)
        r = ((m[2][1] - m[1][2]) * s, (m[0][2] - m[2][0]) * s, (m[1][0] - m[0][1]) * s, f32(0.25) / s)
    elif m[0][0] > m[1][1] and m[0][0] > m[2][2]:
        s = two * np.sqrt(one + m[0][0] - m[1][1] - m[2][2])
        r = (f32(0.25) * s, (m[0][1] + m[1][0]) / s, (m[0][2] + m[2][0]) / s, (m[2][1] - m[1][2]) / s)
    elif m[1][1] > m[2][2]:
        s = two * np.sqrt(one + m[1][1] - m[0][0] - m[2][2])
        r = ((m[0][1] + m[1][0]) / s, f32(0.25) * s, (m[1][2] + m[2][1]) / s, (m[0][2] - m[2][0]) / s)
    else:
        s = two * np.sqrt(one + m[2][2] - m[0][0] - m[1][1])
        r = ((m[0][2] + m[2][0]) / s, (m[1][2] + m[2][1]) / s, f32(0.25) * s, (m[1][0] - m[0][1]) / s)
    return _qnormalized(r)


# ---- the builder (sdtf.rs:124-899) --------------------------------------------------------------------------------------
class _PendingMesh:
    """temp_model's triangle mesh: vertices are appended three per `tri`; the index type widens like
    TriangleIndices::add_index (triangle.rs:267-295)."""

    def __init__(self):
        self.positions, self.normals, self.count = [], [], 0

    def add(self, a, b, c, n):
        self.positions += [a, b, c]
        self.normals += [n, n, n]
        self.count += 3

    def geometry(self) -> TriangleMesh:
        n = self.count
        dtype = np.uint8 if n <= 256 else (np.uint16 if n <= 65536 else np.uint32)   # widened when index 256 / 65536 arrives
        return TriangleMesh(np.array(self.positions, np.float32).reshape(-1, 3), np.arange(n, dtype=dtype),
                            normals=np.array(self.normals, np.float32).reshape(-1, 3))


class SdtfBuilder:
    def __init__(self):
        self._path = self._string = None
        self.vertices = []
        self.transform = []
        self.phong = PhongMaterial(ambient=(0, 0, 0, 1), emission=(0, 0, 0, 1), diffuse=(0, 0, 0, 1), specular=(0, 0, 0, 1), shininess=0.0)
        self.ggx = GgxMaterial(diffuse=(0, 0, 0, 1), specular=(0, 0, 0, 1), roughness=0.0)
        self.attenuation = (1.0, 0.0, 0.0)
        self.pending = None          # _PendingMesh | Sphere
        self.config = SdtfConfig()

    def path(self, p):
        self._path = p
        return self

    def str(self, s):
        self._string = s
        return self

    def build(self):
        model = Model()
        if self._string is not None:
            text = self._string
        elif self._path is not None:
            with open(self._path, "r", encoding="utf-8", newline="") as f:
                text = f.read()
        else:
            raise SdtfError("No path or string provided to load UCSD scene")
        for line in text.split("\n"):
            if line.endswith("\r"):
                line = line[:-1]
            self.parse_line(line, model)
        self.process_primitive(model)
        return model, self.config

    # -- literals ----------------------------------------------------------------------------------------------------------
    @staticmethod
    def _word(w, i, what):
        if i >= len(w):
            raise SdtfError(f"Failed to read {what}")
        return w[i]

    @classmethod
    def _f32(cls, w, i, what):
        s = cls._word(w, i, what).encode()
        end = C.c_char_p()
        v = _libc.strtof(s, C.byref(end))
        if end.value is None or end.value != b"" or not s:
            raise SdtfError(f"invalid float literal `{s.decode()}`")
        return f32(v)

    @classmethod
    def _int(cls, w, i, what):
        s = cls._word(w, i, what)
        try:
            if s.strip() != s or "_" in s:
                raise ValueError
            return int(s, 10)
        except ValueError:
            raise SdtfError(f"invalid digit found in string `{s}`") from None

    @classmethod
    def _u32(cls, w, i, what):
        v = cls._int(w, i, what)
        if not 0 <= v <= 0xFFFFFFFF:
            raise SdtfError(f"number out of range for {what}")
        return v

    @classmethod
    def _vec3(cls, w, i, what):
        return (cls._f32(w, i, what), cls._f32(w, i + 1, what), cls._f32(w, i + 2, what))

    @classmethod
    def _rgb(cls, w, i, what):
        r, g, b = cls._vec3(w, i, what)
        return (float(r), float(g), float(b), 1.0)

    def _current(self):
        if not self.transform:
            raise SdtfError("translate / rotate / scale outside pushTransform (the reference panics: sdtf.rs:357)")
        return len(self.transform) - 1

    # -- one line (sdtf.rs:774-830) ----------------------------------------------------------------------------------------
    def parse_line(self, line, model):
        if line.startswith("#"):
            return
        if not any(ch.isalnum() for ch in line):
            return
        w = [x for x in line.split(" ") if x]
        if not w:
            return
        c = w[0]
        cfg = self.config
        if c == "size":
            cfg.width, cfg.height = self._u32(w, 1, "width"), self._u32(w, 2, "height")
        elif c == "camera":
            self.parse_camera(w, model)
        elif c == "maxverts":
            self._u32(w, 1, "max verts")
        elif c == "vertex":
            self.vertices.append(self._vec3(w, 1, "vertex"))
        elif c == "tri":
            self.parse_tri(w)
        elif c == "ambient":
            self.process_primitive(model)
            self.phong.ambient = self._rgb(w, 1, "ambient")
        elif c == "sphere":
            self.process_primitive(model)
            center, radius = self._vec3(w, 1, "center"), self._f32(w, 4, "radius")
            if self.pending is None:
                self.pending = Sphere(center=tuple(float(x) for x in center), radius=float(radius))
        elif c == "translate":
            i = self._current()
            self.transform[i] = _left_mul(self.transform[i], Trs(translation=tuple(float(x) for x in self._vec3(w, 1, "translation"))))
        elif c == "rotate":
            axis, degrees = self._vec3(w, 1, "rotate"), self._f32(w, 4, "rotate angle")
            i = self._current()
            q = _axis_angle(axis, degrees * (PI32 / f32(180.0)))   # f32::to_radians
            self.transform[i] = _left_mul(self.transform[i], Trs(rotation=tuple(float(x) for x in q)))
        elif c == "scale":
            i = self._current()
            self.transform[i] = _left_mul(self.transform[i], Trs(scale=tuple(float(x) for x in self._vec3(w, 1, "scale"))))
        elif c == "pushTransform":
            self.transform.append(Trs())
        elif c == "popTransform":
            self.process_primitive(model)
            if self.transform:
                self.transform.pop()
        elif c == "emission":
            self.process_primitive(model)
            self.phong.emission = self._rgb(w, 1, "emission")
        elif c == "diffuse":
            self.process_primitive(model)
            self.phong.diffuse = self.ggx.diffuse = self._rgb(w, 1, "diffuse")
        elif c == "specular":
            self.process_primitive(model)
            self.phong.specular = self.ggx.specular = self._rgb(w, 1, "specular")
        elif c == "shininess":
            self.process_primitive(model)
            self.phong.shininess = float(self._f32(w, 1, "shininess"))
        elif c == "roughness":
            self.process_primitive(model)
            self.ggx.roughness = float(self._f32(w, 1, "roughness"))
        elif c == "brdf":
            self.process_primitive(model)
            b = self._word(w, 1, "brdf")
            if b not in ("phong", "ggx"):
                raise SdtfError(f"Failed to find a BRDF for `{b}`")
            cfg.brdf = SdtfBrdfStrategy.Phong if b == "phong" else SdtfBrdfStrategy.Ggx
        elif c == "point":      # sdtf.rs:503-547
            at = self._vec3(w, 1, "point light position")
            light = Light.point()
            light.color, light.attenuation = self._rgb(w, 4, "point light colour"), tuple(float(x) for x in self.attenuation)
            lh = model.lights.push(light)
            model.root.children.append(model.nodes.push(Node(light=lh, trs=Trs(translation=tuple(float(x) for x in at)))))
        elif c == "directional":  # sdtf.rs:549-601
            d = self._vec3(w, 1, "light direction")
            light = Light(kind=abi.LIGHT_DIRECTIONAL, color=self._rgb(w, 4, "directional light colour"), intensity=1.0)
            lh = model.lights.push(light)
            q = _angle_between((1.0, 0.0, 0.0), (-d[0], -d[1], -d[2]))
            model.root.children.append(model.nodes.push(Node(light=lh, trs=Trs(rotation=tuple(float(x) for x in q)))))
        elif c == "attenuation":
            self.attenuation = self._vec3(w, 1, "attenuation")
        elif c == "maxdepth":
            v = self._int(w, 1, "maxdepth")
            if not -2 ** 31 <= v < 2 ** 31:
                raise SdtfError("number out of range for maxdepth")
            cfg.max_depth = v
        elif c == "integrator":
            s = self._word(w, 1, "integrator")
            names = {"raytracer": SdtfIntegratorStrategy.Raytracer, "analyticdirect": SdtfIntegratorStrategy.AnalyticDirect,
                     "direct": SdtfIntegratorStrategy.Direct, "pathtracer": SdtfIntegratorStrategy.Pathtracer}
            if s not in names:
                raise SdtfError(f"Failed to find an integrator for `{s}`")
            cfg.integrator = names[s]
        elif c == "quadLight":  # sdtf.rs:627-706
            a, ab, ac = self._vec3(w, 1, "quad light a"), self._vec3(w, 4, "quad light ab"), self._vec3(w, 7, "quad light ac")
            color = self._rgb(w, 10, "quad light color")
            mh = model.materials.push(PhongMaterial(ambient=(0, 0, 0, 1), emission=color, diffuse=(0, 0, 0, 1), specular=(0, 0, 0, 1), shininess=0.0))
            lh = model.lights.push(Light.quad(ab=tuple(float(x) for x in ab), ac=tuple(float(x) for x in ac), color=color, material=mh, intensity=1.0))
            model.root.children.append(model.nodes.push(Node(light=lh, trs=Trs(translation=tuple(float(x) for x in a)))))
        elif c == "lightsamples":
            cfg.light_samples = self._u32(w, 1, "light samples")
        elif c == "lightstratify":
            cfg.light_stratify = self._word(w, 1, "light_stratify") == "on"
        elif c == "spp":
            cfg.samples_per_pixel = self._u32(w, 1, "spp")
        elif c == "nexteventestimation":
            cfg.direct_sampler = self._sampler(self._word(w, 1, "nexteventestimation"))
        elif c == "russianroulette":
            cfg.russian_roulette = self._word(w, 1, "russianroulette") == "on"
        elif c == "importancesampling":
            cfg.indirect_sampler = self._sampler(self._word(w, 1, "importancesampling"))
        elif c == "gamma":
            cfg.gamma = float(self._f32(w, 1, "gamma"))
        # anything else: "Skipping command" (a log line in the reference)

    @staticmethod
    def _sampler(s):  # sdtf.rs:56-69
        names = {"on": SdtfSamplerStrategy.Nee, "mis": SdtfSamplerStrategy.Mis, "hemisphere": SdtfSamplerStrategy.Hemisphere,
                 "cosine": SdtfSamplerStrategy.Cosine, "brdf": SdtfSamplerStrategy.Brdf}
        if s not in names:
            raise SdtfError(f"Failed to find a sampler for `{s}`")
        return names[s]

    def parse_camera(self, w, model):  # sdtf.rs:173-229
        eye, target, up = self._vec3(w, 1, "camera"), self._vec3(w, 4, "camera target"), self._vec3(w, 7, "camera up")
        yfov_radians = self._f32(w, 10, "camera fov") * PI32 / f32(180.0)
        ch = model.cameras.push(Camera(yfov_radians=float(yfov_radians)))
        rotation = _qinverse(_look_at_rotation(target, eye, up))
        node = Node(camera=ch, trs=Trs(translation=tuple(float(x) for x in eye), rotation=tuple(float(x) for x in rotation)))
        model.root.children.append(model.nodes.push(node))

    def parse_tri(self, w):  # sdtf.rs:247-294
        ia, ib, ic = (self._u32(w, i, "vertex") for i in (1, 2, 3))
        if self.pending is None:
            self.pending = _PendingMesh()
        if not isinstance(self.pending, _PendingMesh):
            return   # the pending primitive is a sphere: the triangle is dropped (`if let` does not match)
        if max(ia, ib, ic) >= len(self.vertices):
            raise SdtfError("tri refers to a vertex that does not exist")
        a, b, c = self.vertices[ia], self.vertices[ib], self.vertices[ic]
        pa, pb, pc = _v(*a, 1.0), _v(*b, 1.0), _v(*c, 1.0)
        n = _normalized(_cross(_sub(pb, pa), _sub(pc, pa)))[:3]
        self.pending.add(a, b, c, n)

    def process_primitive(self, model):  # sdtf.rs:849-870 (+ process_material :833-846)
        if self.pending is None:
            return
        pending, self.pending = self.pending, None
        geometry = pending.geometry() if isinstance(pending, _PendingMesh) else pending
        gh = model.geometries.push(geometry)
        if self.config.brdf == SdtfBrdfStrategy.Phong:
            p = self.phong
            mat = PhongMaterial(ambient=p.ambient, emission=p.emission, diffuse=p.diffuse, specular=p.specular, shininess=p.shininess)
        else:
            g = self.ggx
            mat = GgxMaterial(diffuse=g.diffuse, specular=g.specular, roughness=g.roughness)
        mh = model.materials.push(mat)
        ph = model.primitives.push(Primitive(geometry=gh, material=mh))
        mesh = model.meshes.push(Mesh(primitives=[ph]))
        trs = Trs()
        for t in self.transform:
            trs = _left_mul(trs, t)
        model.root.children.append(model.nodes.push(Node(mesh=mesh, trs=trs)))


def load_sdtf_path(path):
    """Model::load_sdtf_path (sdtf.rs:902-907) -> (Model, SdtfConfig)."""
    return SdtfBuilder().path(path).build()


def load_sdtf_str(text):
    return SdtfBuilder().str(text).build()


def push_sdtf_from_path(scene: Scene, path):
    """Scene::push_sdtf_from_path (rayca-model/src/scene.rs:126-136) -> (scene node handle, SdtfConfig)."""
    model, config = load_sdtf_path(path)
    return scene.push_model(model), config
