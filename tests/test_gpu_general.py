"""GPU parity of the per-pixel stack machine (k_general, rayca_amd/csrc/general.inc): every IntegratorStrategy and
SamplerStrategy of the reference that the generation kernels do not cover, against the CPU oracle on the same
inputs, through the C ABI.  Tolerance as in test_gpu_parity.py: 1e-4 per channel (acos/tan/pow roundings differ
between device libm and glibc), <= 1 LSB after quantisation; a few isolated pixels may amplify such a last-ulp
difference through powf(x, shininess) or a sign test, which the `outliers` bound states per case."""
import numpy as np
import pytest

import oracle_lib as ol
from parity_report import check_outliers
from rayca_amd import (Config, DeviceScene, GgxMaterial, IntegratorStrategy, Light, Mesh, Model, Node, PbrMaterial,
                       PhongMaterial, Primitive, SamplerStrategy, Scene, TriangleMesh, Trs, abi, flatten, scenes)
from rayca_amd.lib import RaycaError

pytestmark = pytest.mark.gpu
TOL = 1e-4
W, H = 96, 72


def quad_light_room(kind="phong"):
    """Room lit by an emissive quad light (the SDTF-style setup: light/quad.rs, nee.rs:72-125, direct.rs)."""
    model = Model()
    if kind == "phong":
        wall = model.materials.push(PhongMaterial(diffuse=(0.7, 0.7, 0.7, 1.0), specular=(0.1, 0.1, 0.1, 1.0), shininess=8.0))
        block = model.materials.push(PhongMaterial(diffuse=(0.2, 0.5, 0.7, 1.0), specular=(0.3, 0.3, 0.3, 1.0), shininess=20.0))
    else:
        wall = model.materials.push(GgxMaterial(diffuse=(0.7, 0.7, 0.7, 1.0), specular=(0.05, 0.05, 0.05, 1.0), roughness=0.8))
        block = model.materials.push(GgxMaterial(diffuse=(0.2, 0.5, 0.7, 1.0), specular=(0.4, 0.4, 0.4, 1.0), roughness=0.35))
    emit = model.materials.push(PhongMaterial(emission=(1.0, 1.0, 1.0, 1.0)))
    X, Y, Z = np.array([2, 0, 0], np.float32), np.array([0, 2, 0], np.float32), np.array([0, 0, 2], np.float32)
    room, blk, lamp = scenes._MeshBuilder(), scenes._MeshBuilder(), scenes._MeshBuilder()
    room.grid((-1, 0, -1), Z, X, 1, 1)
    room.grid((-1, 0, -1), X, Y, 1, 1)
    room.grid((-1, 0, -1), Y, Z, 1, 1)
    room.grid((1, 0, -1), Z, Y, 1, 1)
    blk.box((-0.4, 0, -0.4), (0.2, 0.7, 0.2))
    # the emissive panel the quad light stands for: faces down (-y), like ab x ac below
    lamp.grid((-0.3, 1.9, -0.3), np.array([0.6, 0, 0], np.float32), np.array([0, 0, 0.6], np.float32), 1, 1)
    for mb, mat in ((room, wall), (blk, block), (lamp, emit)):
        g = model.geometries.push(mb.mesh())
        p = model.primitives.push(Primitive(geometry=g, material=mat))
        model.root.children.append(model.nodes.push(Node(mesh=model.meshes.push(Mesh(primitives=[p])))))
    lt = model.lights.push(Light.quad(ab=(0.6, 0.0, 0.0), ac=(0.0, 0.0, 0.6), color=(1, 1, 1, 1), material=emit, intensity=6.0))
    model.root.children.append(model.nodes.push(Node(light=lt, trs=Trs(translation=(-0.3, 1.9, -0.3)))))
    cam = model.cameras.push(scenes.Camera())
    model.root.children.append(model.nodes.push(Node(camera=cam, trs=Trs(translation=(0.0, 1.0, 3.2)))))
    scene = Scene()
    scene.push_model(model)
    return scene


def glass_scene():
    """The glTF box behind a half-transparent pane and over a metallic floor: Scratcher's transmission and
    reflection branches (scratcher.rs:31-43,77-86) and Raytracer's specular recursion both have work to do."""
    scene = scenes.box_scene()
    model = Model()
    pane = model.materials.push(PbrMaterial(color=(0.9, 0.3, 0.2, 0.5), roughness_factor=0.7))
    floor = model.materials.push(PbrMaterial(color=(0.6, 0.6, 0.7, 1.0), metallic_factor=0.8, roughness_factor=0.3))
    quad = model.geometries.push(TriangleMesh.quad())
    pp = model.primitives.push(Primitive(geometry=quad, material=pane))
    fp = model.primitives.push(Primitive(geometry=quad, material=floor))
    model.root.children.append(model.nodes.push(Node(mesh=model.meshes.push(Mesh(primitives=[pp])),
                                                     trs=Trs(translation=(0.3, 0.1, 1.2), scale=(1.2, 1.2, 1.0)))))
    from rayca_amd.model import quat_axis_angle
    model.root.children.append(model.nodes.push(Node(mesh=model.meshes.push(Mesh(primitives=[fp])),
                                                     trs=Trs(translation=(0.0, -0.9, 0.0), rotation=quat_axis_angle((1.0, 0.0, 0.0), -1.5707964),
                                                             scale=(8.0, 8.0, 1.0)))))
    scene.push_model(model)
    return scene


def sphere_scenes(with_boxes):
    import test_gpu_parity as P
    return P._sphere_scene(with_boxes)


def textured_scene():
    """A quad with an albedo + normal texture in front of the box: HitInfo's texture paths under the stack machine."""
    import test_gpu_parity as P
    return P._textured_quad_scene()


SCENES = {
    "sphere": lambda: sphere_scenes(False),
    "sphere_boxes": lambda: sphere_scenes(True),
    "textured": textured_scene,
    "box": scenes.box_scene,
    "cornell": scenes.cornell_scene,
    "room_phong": lambda: quad_light_room("phong"),
    "room_ggx": lambda: quad_light_room("ggx"),
    "glass": glass_scene,
}
_cache = {}


def pair(name):
    if name not in _cache:
        desc = flatten(SCENES[name]())
        _cache[name] = (DeviceScene(desc, Config()), ol.OracleScene(desc, Config()), desc)
    return _cache[name][:2]


I, S = IntegratorStrategy, SamplerStrategy
CASES = [
    # (scene, config, may pixels lie beyond TOL?  The measured count per case is in tests/parity_bounds.json; 0 without entry)
    ("box", Config(integrator=I.Raytracer, max_depth=2), 0.0),
    ("glass", Config(integrator=I.Raytracer, max_depth=3), 0.0),
    ("glass", Config(integrator=I.Scratcher, max_depth=2), 0.0),
    ("box", Config(integrator=I.Scratcher, max_depth=1), 0.0),
    ("room_phong", Config(integrator=I.Direct, light_samples=4, light_stratify=True, seed=5), 0.003),
    ("room_ggx", Config(integrator=I.Direct, light_samples=2, seed=6), 0.0),
    ("room_phong", Config(integrator=I.AnalyticDirect), 0.0),
    ("box", Config(integrator=I.AnalyticDirect), 0.0),
    ("cornell", Config(russian_roulette=True, seed=7), 0.0),
    ("cornell", Config(max_depth=3, light_samples=2, seed=8), 0.0),
    ("room_phong", Config(max_depth=3, light_samples=3, light_stratify=False, seed=9), 0.003),
    ("room_phong", Config(max_depth=2, light_samples=2, direct_sampler=S.Mis, indirect_sampler=S.Brdf, seed=10), 0.003),
    ("room_ggx", Config(max_depth=2, light_samples=2, direct_sampler=S.Mis, indirect_sampler=S.Cosine, seed=11), 0.003),
    ("room_ggx", Config(max_depth=3, indirect_sampler=S.Brdf, seed=12), 0.003),
    ("room_ggx", Config(max_depth=2, direct_sampler=S.NONE, indirect_sampler=S.Hemisphere, seed=13), 0.0),
    ("cornell", Config(max_depth=2, samples_per_pixel=4, light_samples=2, gamma=2.2, seed=14), 0.0),
    # rayca-soft/tests/gltf.rs:10-46 `sphere`: Scratcher without a BVH on the scaled unit sphere
    ("sphere", Config(bvh=False, integrator=I.Scratcher), 0.0),
    ("sphere_boxes", Config(integrator=I.Raytracer, max_depth=2), 0.0),
    ("sphere_boxes", Config(max_depth=3, light_samples=2, seed=15), 0.0),
    ("textured", Config(integrator=I.Raytracer, max_depth=1), 0.0),
    ("textured", Config(russian_roulette=True, indirect_sampler=S.Hemisphere, seed=16), 0.0),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_general_engine_matches_the_oracle(gpu, case):
    name, cfg, outliers = CASES[case]
    ds, orc = pair(name)
    u8, f32, st = ds.render(cfg, W, H, collect_stats=True)
    ou8, of32, ost = orc.render(cfg, W, H)
    assert (f32[..., 3] == 1.0).all()
    assert float(np.nan_to_num(of32[..., :3]).max()) > 0.01, "the case renders nothing"
    # 1e-4 absolute inside the displayable range, relative above it (radiance near a light grows like 1/r^4 and
    # saturates at 255 after quantisation; an absolute bound on a value of 10^3 would ask for more than f32 has)
    n = check_outliers(f"general_case{case:02d}_{name}", f32, of32)
    if outliers == 0.0:
        assert n == 0
        assert np.abs(u8.astype(int) - ou8.astype(int)).max() <= 1
    assert st["rays_shadow"] == ost["rays_shadow"]
    assert st["rays_bounce"] == ost["rays_bounce"]
    assert st["hits_shaded"] == ost["hits_shaded"]


@pytest.mark.parametrize("cfg", [Config(max_depth=3, seed=21), Config(max_depth=1), Config(max_depth=4, indirect_sampler=S.Hemisphere, seed=22),
                                 Config(max_depth=2, samples_per_pixel=4, seed=23)])
def test_general_engine_equals_the_generation_kernels(gpu, cfg):
    """Where both engines apply they must produce the same bits: k_resolve folds the per-depth records in the
    order the recursion evaluates them."""
    ds, _ = pair("cornell")
    _, a, _ = ds.render(cfg, 160, 90)
    _, b, _ = ds.render(cfg, 160, 90, engine=abi.ENGINE_GENERAL)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"max abs diff {np.abs(a - b).max():.3e}"


def test_todo_arms_of_the_reference_are_reported(gpu):
    """Configs that reach a todo!()/panic in the reference on the given scene fail loudly on both sides."""
    for name, cfg in (("room_phong", Config(integrator=I.Raytracer)),             # Light::get_direction of a quad light: todo!()
                      ("box", Config(direct_sampler=S.Mis, max_depth=2)),        # get_t of a Pbr material: todo!()
                      ("room_ggx", Config(integrator=I.Scratcher, max_depth=1))):  # quad light again
        ds, orc = pair(name)
        with pytest.raises(RaycaError) as e:
            ds.render(cfg, 32, 24)
        assert e.value.code == abi.ERR_UNSUPPORTED
        with pytest.raises(ol.OracleError):
            orc.render(cfg, 32, 24)
