"""The oracle must keep reproducing the committed golden vectors (tests/golden/, made by
tests/make_golden.py), and host-side scene logic must be deterministic."""
import os

import numpy as np

import oracle_lib as ol
from rayca_amd import Config, IntegratorStrategy, flatten, scenes

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_box_golden():
    g = np.load(os.path.join(G, "box_256.npz"))
    o = ol.OracleScene(flatten(scenes.box_scene()), Config())
    _, flat, st = o.render(Config(integrator=IntegratorStrategy.Flat), 256, 256)
    assert np.array_equal(_bits(flat), _bits(g["flat"]))
    assert st["rays_primary"] == 65536 and st["hits_shaded"] == 7744
    # the Box is 0.8 red, pbr (www/gltf-model.ts:103-113); flat colour = interpolated white vertex colour x base colour
    lit = flat[flat[..., 0] > 0]
    assert np.all(np.abs(lit[:, 0] - 0.8) < 1e-6) and np.all(lit[:, 1:3] == 0) and np.all(flat[..., 3] == 1)
    _, pt1, st = o.render(Config(max_depth=1), 256, 256)
    assert np.array_equal(_bits(pt1), _bits(g["pt1"]))
    assert st["rays_shadow"] == 2 * 7744  # two point lights (scene.rs:31-52), one shadow ray each per hit
    t, prim, uv, _ = o.trace_rays(g["rays"])
    assert np.array_equal(_bits(t), _bits(g["t"])) and np.array_equal(prim, g["prim"]) and np.array_equal(_bits(uv), _bits(g["uv"]))


def test_cornell_and_soup_golden():
    g = np.load(os.path.join(G, "cornell_128x72.npz"))
    o = ol.OracleScene(flatten(scenes.cornell_scene()), Config())
    _, flat, _ = o.render(Config(integrator=IntegratorStrategy.Flat), 128, 72)
    _, pt1, _ = o.render(Config(max_depth=1), 128, 72)
    assert np.array_equal(_bits(flat), _bits(g["flat"])) and np.array_equal(_bits(pt1), _bits(g["pt1"]))
    t, prim, uv, _ = o.trace_rays(g["rays"])
    assert np.array_equal(_bits(t), _bits(g["t"])) and np.array_equal(prim, g["prim"])
    g = np.load(os.path.join(G, "soup1k_rays.npz"))
    o = ol.OracleScene(flatten(scenes.soup_scene(1000, extent=0.12)), Config())
    assert np.array_equal(o.primitive_order(), g["order"])
    t, prim, uv, _ = o.trace_rays(g["rays"])
    assert np.array_equal(_bits(t), _bits(g["t"])) and np.array_equal(prim, g["prim"]) and np.array_equal(_bits(uv), _bits(g["uv"]))
    assert (prim != 0xFFFFFFFF).sum() > 20


def test_scene_generators_are_deterministic_and_sized():
    a, b = flatten(scenes.soup_scene(5000)), flatten(scenes.soup_scene(5000))
    assert np.array_equal(a.positions, b.positions) and np.array_equal(a.colors, b.colors)
    assert scenes.count_triangles(scenes.cornell_scene()) == 36
    n = scenes.count_triangles(scenes.atrium_scene())
    assert 240_000 <= n <= 290_000, n   # the ~262k-triangle Sponza stand-in


def test_zero_direction_component_pixels_are_black():
    # odd width -> the centre column has dir.x == 0 exactly -> rdir.x = 0 -> every box missed
    o = ol.OracleScene(flatten(scenes.box_scene()), Config())
    _, f, _ = o.render(Config(integrator=IntegratorStrategy.Flat), 65, 64)
    assert np.all(f[:, 32, :3] == 0) and f[32, 31, 0] > 0 and f[32, 33, 0] > 0


def test_oracle_error_paths():
    from rayca_amd import Model, Scene, abi
    import pytest
    scene = Scene()
    scene.push_model(scenes.load_gltf(os.path.join(G, "box.gltf")))  # no camera
    o = ol.OracleScene(flatten(scene), Config())
    with pytest.raises(ol.OracleError) as e:
        o.render(Config(), 8, 8)
    assert e.value.code == abi.ERR_NO_CAMERA
    scene = Scene()
    scene.push_model(scenes.create_default_model())  # camera + lights, no geometry
    o = ol.OracleScene(flatten(scene), Config())
    with pytest.raises(ol.OracleError) as e:
        o.render(Config(), 8, 8)
    assert e.value.code == abi.ERR_EMPTY_SCENE
