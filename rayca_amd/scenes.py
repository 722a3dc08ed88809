"""Synthetic workloads for the hot path (BASELINE.json `configs`), built with the host mirror.

  box_scene      C1  the Khronos Box glTF (12 triangles) + create_default_model(), as the reference's
                     `gltf::cube` test does (rayca-soft/tests/gltf.rs:191-204)
  cornell_scene  C2  Cornell-box-style room: 5 walls + 2 boxes + ceiling panel (36 triangles)
  atrium_scene   C3/C5  procedural colonnaded atrium, ~262k triangles.  A STAND-IN for Sponza: the
                     real Sponza asset (tests/model, an un-vendored submodule of the reference) is not
                     available offline.  Named as such in every report.
  soup_scene     C4  1,048,576 random triangles, seed 0x5EED0001

All generators are deterministic functions of their arguments (integer hashing + f32 arithmetic in
numpy), so the CPU container and the GPU box build bit-identical scenes.
"""
from __future__ import annotations

import math
import os

import numpy as np

from . import abi
from .gltf import load_gltf
from .model import (Camera, Light, Mesh, Model, Node, PbrMaterial, Primitive, Scene, TriangleMesh, Trs,
                    create_default_model, quat_axis_angle)

_GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


# ---- deterministic hashing -----------------------------------------------------------------------
def hash_u32(seed: int, idx: np.ndarray) -> np.ndarray:
    """murmur3-finaliser based counter hash, vectorised (uint32 wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        h = (idx.astype(np.uint32) * np.uint32(0x9E3779B1)) + np.uint32(seed & 0xFFFFFFFF)
        for _ in range(2):
            h ^= h >> np.uint32(16)
            h *= np.uint32(0x85EBCA6B)
            h ^= h >> np.uint32(13)
            h *= np.uint32(0xC2B2AE35)
            h ^= h >> np.uint32(16)
    return h


def hash_unit(seed: int, idx: np.ndarray) -> np.ndarray:
    """uniform f32 in [0,1): 23 mantissa bits, like the renderer's RNG"""
    u = hash_u32(seed, idx)
    bits = np.uint32(0x3F800000) | (u >> np.uint32(9))
    return bits.view(np.float32) - np.float32(1.0)


# ---- helpers -------------------------------------------------------------------------------------
class _MeshBuilder:
    """Accumulates non-shared-vertex geometry for one material."""

    def __init__(self):
        self.pos, self.nrm, self.col, self.idx = [], [], [], []
        self.nv = 0

    def add(self, pos, nrm, col, idx):
        pos = np.asarray(pos, np.float32).reshape(-1, 3)
        self.pos.append(pos)
        self.nrm.append(np.asarray(nrm, np.float32).reshape(-1, 3))
        self.col.append(np.broadcast_to(np.asarray(col, np.float32), (pos.shape[0], 4)).copy())
        self.idx.append(np.asarray(idx, np.uint32).reshape(-1) + np.uint32(self.nv))
        self.nv += pos.shape[0]

    def grid(self, origin, u, v, nu, nv, col=(1, 1, 1, 1), displace=None):
        """(nu x nv) quads spanning origin + s*u + t*v; front face (normal) = u x v."""
        origin, u, v = (np.asarray(a, np.float32) for a in (origin, u, v))
        s = np.linspace(0, 1, nu + 1, dtype=np.float32)
        t = np.linspace(0, 1, nv + 1, dtype=np.float32)
        S, T = np.meshgrid(s, t, indexing="ij")
        P = origin[None, None, :] + S[..., None] * u[None, None, :] + T[..., None] * v[None, None, :]
        n = np.cross(u, v)
        n = (n / np.linalg.norm(n)).astype(np.float32)
        if displace is not None:
            P = P + displace(S, T)[..., None] * n[None, None, :]
        P = P.astype(np.float32)
        i0 = (np.arange(nu)[:, None] * (nv + 1) + np.arange(nv)[None, :]).reshape(-1)
        i1, i2, i3 = i0 + (nv + 1), i0 + (nv + 1) + 1, i0 + 1  # +u, +u+v, +v
        idx = np.stack([i0, i1, i2, i0, i2, i3], 1).reshape(-1)
        self.add(P.reshape(-1, 3), np.tile(n, ((nu + 1) * (nv + 1), 1)), col, idx)

    def box(self, lo, hi, col=(1, 1, 1, 1), tess=1):
        """axis-aligned box, outward faces"""
        lo, hi = np.asarray(lo, np.float32), np.asarray(hi, np.float32)
        d = hi - lo
        X, Y, Z = np.array([d[0], 0, 0], np.float32), np.array([0, d[1], 0], np.float32), np.array([0, 0, d[2]], np.float32)
        self.grid(lo + Z, X, Y, tess, tess, col)            # +z face: x cross y = +z
        self.grid(lo, Y, X, tess, tess, col)                # -z
        self.grid(lo + X, Y, Z, tess, tess, col)            # +x: y cross z = +x
        self.grid(lo, Z, Y, tess, tess, col)                # -x
        self.grid(lo + Y, Z, X, tess, tess, col)            # +y: z cross x = +y
        self.grid(lo, X, Z, tess, tess, col)                # -y

    def cylinder(self, base, radius, height, sides, segments, col=(1, 1, 1, 1), flute=0.0):
        """vertical cylinder shell, outward facing; optional fluting (radial ripple)"""
        base = np.asarray(base, np.float32)
        a = np.linspace(0, 2 * math.pi, sides + 1, dtype=np.float32)
        h = np.linspace(0, 1, segments + 1, dtype=np.float32)
        A, H = np.meshgrid(a, h, indexing="ij")
        r = np.float32(radius) * (1 + np.float32(flute) * np.cos(A * 12).astype(np.float32))
        P = np.stack([base[0] + r * np.cos(A), base[1] + H * np.float32(height), base[2] + r * np.sin(A)], -1).astype(np.float32)
        N = np.stack([np.cos(A), np.zeros_like(A), np.sin(A)], -1).astype(np.float32)
        i0 = (np.arange(sides)[:, None] * (segments + 1) + np.arange(segments)[None, :]).reshape(-1)
        i1, i2, i3 = i0 + (segments + 1), i0 + (segments + 1) + 1, i0 + 1
        # +angle then +height: d/dA x d/dH = (-sin,0,cos) x (0,1,0) = (-cos, 0, -sin) -> inward; flip
        idx = np.stack([i0, i2, i1, i0, i3, i2], 1).reshape(-1)
        self.add(P.reshape(-1, 3), N.reshape(-1, 3), col, idx)

    def mesh(self) -> TriangleMesh:
        return TriangleMesh(np.concatenate(self.pos), np.concatenate(self.idx), colors=np.concatenate(self.col),
                            normals=np.concatenate(self.nrm))

    def triangles(self) -> int:
        return sum(i.size for i in self.idx) // 3


def _single_model_scene(parts, camera_trs: Trs, yfov, lights=()):
    """parts: list of (TriangleMesh, material).  One model, one mesh node per part, + camera/lights."""
    model = Model("synthetic")
    for tm, mat in parts:
        mh = model.materials.push(mat)
        g = model.geometries.push(tm)
        p = model.primitives.push(Primitive(geometry=g, material=mh))
        m = model.meshes.push(Mesh(primitives=[p]))
        n = model.nodes.push(Node(mesh=m))
        model.root.children.append(n)
    cam = model.cameras.push(Camera(yfov_radians=yfov))
    model.root.children.append(model.nodes.push(Node(camera=cam, trs=camera_trs)))
    for pos, intensity in lights:
        lt = Light.point()
        lt.set_intensity(intensity)
        lh = model.lights.push(lt)
        model.root.children.append(model.nodes.push(Node(light=lh, trs=Trs(translation=tuple(pos)))))
    scene = Scene("synthetic")
    scene.push_model(model)
    return scene


# ---- C1 ------------------------------------------------------------------------------------------
def box_scene(path=None) -> Scene:
    """rayca-soft/tests/gltf.rs:191-204 `gltf::cube`: box.gltf + create_default_model()."""
    scene = Scene("box")
    scene.push_model(load_gltf(path or os.path.join(_GOLDEN, "box.gltf")))
    scene.push_model(create_default_model())
    return scene


def triangle_scene() -> Scene:
    """rayca-soft/tests/gltf.rs:48-84 `triangle`: coloured unit triangle, scaled and translated."""
    model = Model()
    tri = TriangleMesh.unit()
    tri.colors = np.array([[1, 0, 0, 1], [0, 1, 0, 1], [0, 0, 1, 1]], np.float32)  # Color::from(0xRRGGBBAA)
    g = model.geometries.push(tri)
    p = model.primitives.push(Primitive(geometry=g))
    m = model.meshes.push(Mesh(primitives=[p]))
    n = model.nodes.push(Node(mesh=m, trs=Trs(translation=(0.0, -1.0, 0.0), scale=(1.0, 2.0, 1.0))))
    model.root.children.append(n)
    scene = Scene("triangle")
    scene.push_model(model)
    scene.push_model(create_default_model())
    return scene


# ---- C2 ------------------------------------------------------------------------------------------
def cornell_scene(with_light=True) -> Scene:
    """Cornell-box-style room x,z in [-1,1], y in [0,2]; camera at (0,1,3.4) looking down -Z, yfov 45 deg."""
    white, red, green = (0.73, 0.73, 0.73, 1), (0.65, 0.05, 0.05, 1), (0.12, 0.45, 0.15, 1)
    X, Y, Z = np.array([2, 0, 0], np.float32), np.array([0, 2, 0], np.float32), np.array([0, 0, 2], np.float32)
    w, r, g, b = _MeshBuilder(), _MeshBuilder(), _MeshBuilder(), _MeshBuilder()
    w.grid((-1, 0, -1), Z, X, 1, 1)           # floor, normal +y
    w.grid((-1, 2, -1), X, Z, 1, 1)           # ceiling, normal -y
    w.grid((-1, 0, -1), X, Y, 1, 1)           # back wall, normal +z
    r.grid((-1, 0, -1), Y, Z, 1, 1)           # left wall (red), normal +x
    g.grid((1, 0, -1), Z, Y, 1, 1)            # right wall (green), normal -x
    w.grid((-0.3, 1.995, -0.3), np.array([0.6, 0, 0], np.float32), np.array([0, 0, 0.6], np.float32), 1, 1)  # ceiling panel
    b.box((-0.75, 0, -0.65), (-0.15, 1.2, -0.05))   # tall box
    b.box((0.1, 0, 0.0), (0.7, 0.6, 0.6))           # short box
    parts = [(w.mesh(), PbrMaterial(color=white, roughness_factor=1.0)),
             (r.mesh(), PbrMaterial(color=red, roughness_factor=1.0)),
             (g.mesh(), PbrMaterial(color=green, roughness_factor=1.0)),
             (b.mesh(), PbrMaterial(color=white, roughness_factor=0.6))]
    lights = [((0.0, 1.9, 0.0), 4.0)] if with_light else []
    return _single_model_scene(parts, Trs(translation=(0.0, 1.0, 3.4)), math.pi / 4, lights)


# ---- C4 ------------------------------------------------------------------------------------------
def soup_scene(n_tris: int = 1 << 20, seed: int = 0x5EED0001, extent: float = 0.01) -> Scene:
    """n_tris random triangles: centres uniform in [-1,1]^3, each corner = centre + offset uniform in
    [-extent,extent]^3; one random colour per triangle; camera at (0,0,3.5), yfov pi/4."""
    i = np.arange(n_tris * 3, dtype=np.uint32)
    f = np.float32
    centre = (hash_unit(seed + 1, i).reshape(-1, 3) * f(2) - f(1)).astype(np.float32)
    off = (hash_unit(seed + 2, np.arange(n_tris * 9, dtype=np.uint32)).reshape(-1, 3, 3) * f(2) - f(1)) * f(extent)
    pos = (centre[:, None, :] + off).astype(np.float32).reshape(-1, 3)
    rgb = hash_unit(seed + 3, i).reshape(-1, 3) * f(0.8) + f(0.2)
    col = np.concatenate([rgb, np.ones((n_tris, 1), np.float32)], 1)
    col = np.repeat(col, 3, axis=0)
    idx = np.arange(n_tris * 3, dtype=np.uint32)
    tm = TriangleMesh(pos, idx, colors=col)
    return _single_model_scene([(tm, PbrMaterial(color=(1, 1, 1, 1), roughness_factor=1.0))],
                               Trs(translation=(0.0, 0.0, 3.5)), math.pi / 4, [((0.0, 3.0, 3.0), 64.0)])


# ---- C3 / C5 -------------------------------------------------------------------------------------
def atrium_scene(detail: int = 14) -> Scene:
    """Procedural colonnaded atrium (Sponza STAND-IN).  detail=14 gives ~272k triangles (Sponza has ~262k): a closed hall
    30 x 10 x 12 with displaced stone walls, two colonnades of fluted columns carrying an upper
    gallery with a balustrade, hanging drapes and floor tiles.  One point light under the roof; the
    camera stands at one end looking down the nave (like the reference's ignored `sponza` test,
    rayca-soft/tests/gltf.rs:296-342)."""
    d = max(1, int(detail))
    L, W, H = 15.0, 6.0, 10.0  # half length (x), half width (z), height (y)
    stone, floorc, colc, drape_r, drape_g, drape_b = (0.62, 0.58, 0.5, 1), (0.45, 0.42, 0.4, 1), (0.7, 0.68, 0.62, 1), (0.6, 0.1, 0.1, 1), (0.1, 0.45, 0.15, 1), (0.1, 0.15, 0.55, 1)
    walls, floor, cols, dr, dg, db = (_MeshBuilder() for _ in range(6))

    def bricks(amp, fu, fv):
        def fn(S, T):
            return (np.float32(amp) * (np.sin(S * np.float32(fu)) * np.sin(T * np.float32(fv)))).astype(np.float32)
        return fn

    X2, Z2, Y = np.array([2 * L, 0, 0], np.float32), np.array([0, 0, 2 * W], np.float32), np.array([0, H, 0], np.float32)
    floor.grid((-L, 0, -W), Z2, X2, 6 * d, 15 * d, floorc)                                   # floor (+y)
    walls.grid((-L, H, -W), X2, Z2, 4 * d, 2 * d, stone)                                      # roof (-y)
    walls.grid((-L, 0, -W), X2, Y, 12 * d, 5 * d, stone, bricks(0.04, 90.0, 40.0))            # z=-W wall (+z)
    walls.grid((-L, 0, W), Y, X2, 5 * d, 12 * d, stone, bricks(0.04, 40.0, 90.0))             # z=+W wall (-z)
    walls.grid((-L, 0, -W), Y, Z2, 4 * d, 5 * d, stone, bricks(0.03, 30.0, 40.0))             # x=-L wall (+x)
    walls.grid((L, 0, -W), Z2, Y, 5 * d, 4 * d, stone, bricks(0.03, 40.0, 30.0))              # x=+L wall (-x)
    # colonnades + gallery slabs
    ncol = 10
    for side in (-1.0, 1.0):
        z = side * 3.6
        for k in range(ncol):
            x = -L + 1.5 + k * (2 * L - 3.0) / (ncol - 1)
            cols.box((x - 0.45, 0, z - 0.45), (x + 0.45, 0.35, z + 0.45), colc, 1)                 # plinth
            cols.cylinder((x, 0.35, z), 0.3, 4.0, 3 * d, 2 * d, colc, flute=0.04)                 # shaft
            cols.box((x - 0.45, 4.35, z - 0.45), (x + 0.45, 4.7, z + 0.45), colc, 1)               # capital
            cols.cylinder((x, 5.3, z), 0.2, 3.2, 2 * d, d, colc, flute=0.03)                      # upper shaft
        lo_z, hi_z = (z - 0.6, W) if side > 0 else (-W, z + 0.6)
        walls.box((-L, 4.7, lo_z), (L, 5.0, hi_z), stone, 2 * d)                                   # gallery slab
        # balustrade: small posts + rail
        nb = 30 * d // 4 + 2
        zr = z - side * 0.45
        for k in range(nb):
            x = -L + 0.5 + k * (2 * L - 1.0) / (nb - 1)
            cols.cylinder((x, 5.0, zr), 0.05, 0.8, 6, 1, colc)
        walls.box((-L + 0.3, 5.8, zr - 0.08), (L - 0.3, 5.9, zr + 0.08), stone, max(1, d // 2))
    # hanging drapes between columns (double sided: back faces are culled by the renderer)
    drapes = [dr, dg, db]
    nd = 9
    for side in (-1.0, 1.0):
        z = side * 3.6
        for k in range(nd):
            x0 = -L + 1.5 + k * (2 * L - 3.0) / (ncol - 1) + 0.5
            wdt = (2 * L - 3.0) / (ncol - 1) - 1.0
            mb = drapes[k % 3]
            colr = (drape_r, drape_g, drape_b)[k % 3]
            wave = bricks(0.12, 25.0, 3.0)
            U, V = np.array([wdt, 0, 0], np.float32), np.array([0, -2.6, 0], np.float32)
            mb.grid((x0, 4.6, z), U, V, 3 * d, 2 * d, colr, wave)       # front (normal -z ... u x v = (w,0,0)x(0,-h,0) = -z)
            mb.grid((x0, 4.6, z + 0.02), V, U, 2 * d, 3 * d, colr,
                    lambda S, T, wv=wave: -wv(T, S))                      # back (+z), same surface shape
    parts = [(walls.mesh(), PbrMaterial(color=(1, 1, 1, 1), roughness_factor=0.9)),
             (floor.mesh(), PbrMaterial(color=(1, 1, 1, 1), roughness_factor=0.5, metallic_factor=0.1)),
             (cols.mesh(), PbrMaterial(color=(1, 1, 1, 1), roughness_factor=0.7)),
             (dr.mesh(), PbrMaterial(color=(1, 1, 1, 1), roughness_factor=1.0)),
             (dg.mesh(), PbrMaterial(color=(1, 1, 1, 1), roughness_factor=1.0)),
             (db.mesh(), PbrMaterial(color=(1, 1, 1, 1), roughness_factor=1.0))]
    # camera at the -x end, looking toward +x: default camera looks down -Z, rotate -90 deg about Y
    cam_trs = Trs(translation=(-L + 1.0, 2.2, 0.3), rotation=quat_axis_angle((0.0, 1.0, 0.0), -math.pi / 2))
    return _single_model_scene(parts, cam_trs, math.radians(60.0), [((2.0, 6.5, 0.5), 40000.0)])


def count_triangles(scene: Scene) -> int:
    n = 0
    for m in scene.models:
        for g in m.geometries:
            if isinstance(g, TriangleMesh):
                n += int(np.asarray(g.indices).size) // 3
    return n


WORKLOADS = {
    "box": dict(scene=box_scene, width=256, height=256),
    "cornell": dict(scene=cornell_scene, width=1920, height=1080),
    "atrium": dict(scene=atrium_scene, width=1920, height=1080),
    "soup": dict(scene=soup_scene, width=4096, height=4096),
}
