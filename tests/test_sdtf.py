"""SDTF loader + Config::apply: the C++ loader (include/rayca_sdtf.hpp) and the Python loader
(rayca_amd/sdtf.py) against the reference's own unit tests (rayca-model/src/loader/sdtf.rs:913-947), against each other
(same flattened bytes, same Config), and -- on the GPU box -- the HIP path against the oracle on the loaded scenes."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
from rayca_amd import Config, IntegratorStrategy, SamplerStrategy, abi, flatten, model as M, sdtf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
FIXTURES = ["cornell_quad.sdtf", "spheres.sdtf"]


@pytest.fixture(scope="module")
def host_mirror(product_lib):
    import __graft_entry__ as g
    return g.build_cpp_host()


def scene_of(path):
    """rayca-soft/tests/sdtf.rs:7-13: Scene::default() + push_sdtf_from_path."""
    scene = M.Scene()
    _, cfg = sdtf.push_sdtf_from_path(scene, path)
    return scene, cfg


# ---- the reference's unit tests, restated for both loaders --------------------------------------------------------------
def test_size():  # sdtf.rs:913-919
    _, cfg = sdtf.load_sdtf_str("size 320 240")
    assert (cfg.width, cfg.height) == (320, 240)


def test_camera():  # sdtf.rs:921-930
    model, _ = sdtf.load_sdtf_str("camera -4 -4 4 1 0 0 0 1 0 30")
    assert np.float32(model.cameras[0].yfov_radians) == np.float32(30.0) * (np.float32(np.pi) / np.float32(180.0))   # 30.0f32.to_radians()


TRIANGLE = """
maxverts 3
vertex -1 -1 0
vertex +1 -1 0
vertex +1 +1 0
tri 0 1 2"""


def test_triangle():  # sdtf.rs:932-947
    model, _ = sdtf.load_sdtf_str(TRIANGLE)
    tri = model.geometries[model.primitives[0].geometry]
    assert isinstance(tri, M.TriangleMesh) and tri.positions.shape == (3, 3)
    assert np.array_equal(tri.normals, np.tile(np.array([[0, 0, 1]], np.float32), (3, 1)))   # cross(ab, ac) normalised


def test_reference_unit_tests_through_the_cpp_loader(host_mirror, tmp_path):
    subprocess.run([host_mirror, "sdtf_config", "str:size 320 240", str(tmp_path)], check=True)
    v = np.frombuffer((tmp_path / "sdtf_config.bin").read_bytes(), np.int64)
    assert (v[0], v[1]) == (320, 240)
    subprocess.run([host_mirror, "describe", "sdtfstr:camera -4 -4 4 1 0 0 0 1 0 30", str(tmp_path)], check=True)
    cams = np.frombuffer((tmp_path / "cameras.bin").read_bytes(), np.float32)
    assert cams[0] == np.float32(30.0) * (np.float32(np.pi) / np.float32(180.0))
    subprocess.run([host_mirror, "describe", "sdtfstr:" + TRIANGLE, str(tmp_path)], check=True)
    assert np.frombuffer((tmp_path / "positions.bin").read_bytes(), np.float32).size == 9   # triangles.vertices.len() == 3


# ---- loader behaviour the reference has and a tidy parser would not -----------------------------------------------------
def test_reference_quirks_are_kept():
    model, cfg = sdtf.load_sdtf_path(os.path.join(G, "spheres.sdtf"))
    kinds = [type(model.geometries[p.geometry]).__name__ for p in model.primitives]
    assert kinds == ["TriangleMesh", "Sphere", "Sphere", "Sphere"]
    assert model.geometries[0].positions.shape[0] == 6          # the `tri` behind the first sphere was dropped (sdtf.rs:262-289)
    assert len(model.materials) == 4                             # one material copy per primitive (sdtf.rs:833-848)
    assert [tuple(l.attenuation) for l in model.lights] == [(0.0, 0.0, 1.0), (1.0, np.float32(0.05), np.float32(0.02))]
    assert cfg.integrator == sdtf.SdtfIntegratorStrategy.Raytracer and cfg.direct_sampler == sdtf.SdtfSamplerStrategy.NONE
    with pytest.raises(sdtf.SdtfError):
        sdtf.load_sdtf_str("translate 1 2 3")                    # no pushTransform: `last_mut().unwrap()` panics (sdtf.rs:357)
    with pytest.raises(sdtf.SdtfError):
        sdtf.load_sdtf_str("integrator whitted")
    with pytest.raises(sdtf.SdtfError):
        sdtf.load_sdtf_str("vertex 1 2")                         # .expect("Failed to read vertex z")
    m, _ = sdtf.load_sdtf_str("# comment\n   \n\tsize\t9 9\nsize  7   5")   # tabs do not separate words (sdtf.rs:786)
    assert _.width == 7 and _.height == 5


def test_config_apply():  # rayca-soft/src/config.rs:58-71
    _, s = sdtf.load_sdtf_str("maxdepth -1\nlightsamples 9\nlightstratify on\nspp 4\nnexteventestimation mis\nimportancesampling brdf\n"
                              "integrator direct\ngamma 2.2\nrussianroulette on")
    c = sdtf.apply(Config(bvh=False), s)
    assert c.max_depth == 16 and c.light_samples == 9 and c.light_stratify and c.samples_per_pixel == 4
    assert c.direct_sampler == SamplerStrategy.Mis and c.indirect_sampler == SamplerStrategy.Brdf and c.integrator == IntegratorStrategy.Direct
    assert np.float32(c.gamma) == np.float32(2.2)
    assert c.bvh is False and c.russian_roulette is False        # neither is copied by Config::apply
    assert s.russian_roulette is True
    d = sdtf.apply(Config(), sdtf.SdtfConfig())                  # SdtfConfig::default(): Raytracer, no NEE, hemisphere
    assert (d.integrator, d.direct_sampler, d.indirect_sampler, d.max_depth) == (IntegratorStrategy.Raytracer, SamplerStrategy.NONE, SamplerStrategy.Hemisphere, 5)


# ---- C++ loader == Python loader ------------------------------------------------------------------------------------------
def _raw(ctypes_array, count, ctype):
    return bytes(C.string_at(ctypes_array, count * C.sizeof(ctype))) if count else b""


@pytest.mark.parametrize("name", FIXTURES)
def test_cpp_and_python_loaders_flatten_to_the_same_bytes(host_mirror, tmp_path, name):
    path = os.path.join(G, name)
    subprocess.run([host_mirror, "describe", "sdtf:" + path, str(tmp_path)], check=True)
    subprocess.run([host_mirror, "sdtf_config", path, str(tmp_path)], check=True)
    scene, scfg = scene_of(path)
    d = flatten(scene)
    c = d.c

    def blob(n):
        return (tmp_path / f"{n}.bin").read_bytes()

    assert blob("nodes") == _raw(c.nodes, c.node_count, abi.RaycaNode)           # every Trs::left_mul product, bit for bit
    assert blob("meshes") == _raw(c.meshes, c.mesh_count, abi.RaycaMesh)
    assert blob("primitives") == _raw(c.primitives, c.primitive_count, abi.RaycaPrimitive)
    assert blob("materials") == _raw(c.materials, c.material_count, abi.RaycaMaterial)
    assert blob("cameras") == _raw(c.cameras, c.camera_count, abi.RaycaCamera)
    assert blob("lights") == _raw(c.lights, c.light_count, abi.RaycaLight)
    assert blob("index_bytes") == d.index_bytes.tobytes()
    assert blob("positions") == d.positions.tobytes()
    assert blob("normals") == d.normals.tobytes()
    cfg = sdtf.apply(Config(bvh=False), scfg)
    want = [scfg.width, scfg.height, scfg.max_depth, scfg.light_samples, int(scfg.light_stratify), scfg.samples_per_pixel, scfg.direct_sampler,
            int(scfg.russian_roulette), scfg.indirect_sampler, scfg.integrator, scfg.brdf, int(np.float32(scfg.gamma).view(np.uint32)),
            int(cfg.bvh), cfg.light_samples, int(cfg.light_stratify), cfg.samples_per_pixel, int(cfg.russian_roulette), cfg.direct_sampler,
            cfg.indirect_sampler, cfg.integrator, cfg.max_depth, int(np.float32(cfg.gamma).view(np.uint32))]
    assert list(np.frombuffer(blob("sdtf_config"), np.int64)) == want


def test_index_type_widens_like_add_index():
    """TriangleIndices::add_index (triangle.rs:267-295): u8 until index 256 arrives, then u16."""
    text = "maxverts 3\nvertex 0 0 0\nvertex 1 0 0\nvertex 0 1 0\n" + "tri 0 1 2\n" * 86
    model, _ = sdtf.load_sdtf_str(text)
    assert model.geometries[0].indices.dtype == np.uint16 and model.geometries[0].indices.size == 258
    model, _ = sdtf.load_sdtf_str("maxverts 3\nvertex 0 0 0\nvertex 1 0 0\nvertex 0 1 0\n" + "tri 0 1 2\n" * 85)
    assert model.geometries[0].indices.dtype == np.uint8 and model.geometries[0].indices.size == 255


def test_cpp_index_type_widens_too(host_mirror, tmp_path):
    text = "maxverts 3\nvertex 0 0 0\nvertex 1 0 0\nvertex 0 1 0\n" + "tri 0 1 2\n" * 86
    subprocess.run([host_mirror, "describe", "sdtfstr:" + text, str(tmp_path)], check=True)
    prim = np.frombuffer((tmp_path / "primitives.bin").read_bytes(), np.uint8)
    p = abi.RaycaPrimitive.from_buffer_copy(prim.tobytes()[: C.sizeof(abi.RaycaPrimitive)])
    assert p.index_type == abi.INDEX_U16 and p.index_count == 258
    scene = M.Scene()
    scene.push_model(sdtf.load_sdtf_str(text)[0])
    assert (tmp_path / "index_bytes.bin").read_bytes() == flatten(scene).index_bytes.tobytes()


# ---- the oracle renders what the loader produces (plumbing, CPU) ----------------------------------------------------------
@pytest.mark.parametrize("name", FIXTURES)
def test_oracle_renders_the_loaded_scene(name):
    scene, scfg = scene_of(os.path.join(G, name))
    cfg = sdtf.apply(Config(bvh=False), scfg)       # rayca-soft/tests/sdtf.rs: most SDTF tests force bvh(false)
    orc = ol.OracleScene(flatten(scene), cfg, threads=4)
    u8, f32, st = orc.render(cfg, scfg.width // 4, scfg.height // 4)
    assert (f32[..., 3] == 1.0).all() and float(f32[..., :3].max()) > 0.05
    assert st["hits_shaded"] > 100


# ---- GPU == oracle on the loaded scenes ------------------------------------------------------------------------------------
I, S = IntegratorStrategy, SamplerStrategy
GPU_CASES = [
    ("cornell_quad.sdtf", None),                                                     # the file's own Config: Pathtracer depth 3, NEE 4 stratified
    ("cornell_quad.sdtf", dict(integrator=I.Direct)),
    ("cornell_quad.sdtf", dict(integrator=I.AnalyticDirect)),
    ("cornell_quad.sdtf", dict(direct_sampler=S.Mis, indirect_sampler=S.Brdf, max_depth=2, light_samples=2, light_stratify=False)),
    ("spheres.sdtf", dict(integrator=I.Pathtracer, direct_sampler=S.Nee, indirect_sampler=S.Cosine, max_depth=3)),
    ("spheres.sdtf", dict(integrator=I.Pathtracer, direct_sampler=S.Nee, indirect_sampler=S.Cosine, max_depth=1)),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(len(GPU_CASES)))
def test_gpu_matches_the_oracle_on_sdtf_scenes(gpu, case):
    from parity_report import check_outliers
    from rayca_amd import DeviceScene
    name, override = GPU_CASES[case]
    scene, scfg = scene_of(os.path.join(G, name))
    desc = flatten(scene)
    for bvh in (False, True):
        cfg = sdtf.apply(Config(bvh=bvh, seed=40 + case), scfg)
        for k, v in (override or {}).items():
            setattr(cfg, k, v)
        ds = DeviceScene(desc, cfg)
        orc = ol.OracleScene(desc, cfg)
        u8, f32, st = ds.render(cfg, scfg.width, scfg.height, collect_stats=True)
        ou8, of32, ost = orc.render(cfg, scfg.width, scfg.height)
        assert float(np.nan_to_num(of32[..., :3]).max()) > 0.01, "the case renders nothing"
        check_outliers(f"sdtf_case{case}_{name.split('.')[0]}_bvh{int(bvh)}", f32, of32)
        assert st["rays_shadow"] == ost["rays_shadow"] and st["hits_shaded"] == ost["hits_shaded"]
        ds.close()
        orc.close()


@pytest.mark.gpu
def test_gpu_whitted_raytracer_on_the_sphere_scene(gpu):
    """spheres.sdtf as the file asks: the Raytracer integrator, point lights with two attenuation settings."""
    from parity_report import check_outliers
    from rayca_amd import DeviceScene
    scene, scfg = scene_of(os.path.join(G, "spheres.sdtf"))
    desc = flatten(scene)
    cfg = sdtf.apply(Config(bvh=False), scfg)
    assert cfg.integrator == I.Raytracer and cfg.max_depth == 2
    ds, orc = DeviceScene(desc, cfg), ol.OracleScene(desc, cfg)
    u8, f32, st = ds.render(cfg, scfg.width, scfg.height, collect_stats=True)
    ou8, of32, ost = orc.render(cfg, scfg.width, scfg.height)
    check_outliers("sdtf_spheres_raytracer", f32, of32)
    assert int(np.abs(u8.astype(int) - ou8.astype(int)).max()) <= 1
    assert float(f32[..., :3].max()) > 0.2
