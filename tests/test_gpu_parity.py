"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle on the same inputs.

Bars: hit records (t, primitive, u, v) and every Flat pixel are bit-exact (pure +,-,*,/,sqrt f32
arithmetic in the reference's operation order); shaded pixels go through the GGX terms
(brdf/ggx.rs:58-83), which the oracle spells with the reference's acos/tan/pow compositions and the kernels
evaluate in closed form (kernels.hip, RAYCA_GGX_CLOSED_FORM): |gpu - oracle| <= 1e-4 per channel (BASELINE.json
north_star) inside the displayable range [0, 1], 1e-4 relative above it (radiance next to a light reaches 10^3,
where one f32 ulp is already 1.2e-4), and <= 1 LSB after RGBA8 quantisation.
"""
import os

import numpy as np
import pytest

import oracle_lib as ol
from parity_report import check_outliers, check_u8_outliers
from rayca_amd import (Config, DeviceScene, Image, IntegratorStrategy, Light, Mesh, Model, Node, PbrMaterial,
                       PhongMaterial, Primitive, SamplerStrategy, Scene, SoftRenderer, Texture, TriangleMesh, Trs, abi,
                       flatten, scenes)
from rayca_amd.lib import RaycaError

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-4
FLAT = Config(integrator=IntegratorStrategy.Flat)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def pair(desc, cfg=None, **okw):
    return DeviceScene(desc, cfg or Config()), ol.OracleScene(desc, cfg or Config(), **okw)


def assert_exact(gpu_f32, ora_f32):
    assert np.array_equal(bits(gpu_f32), bits(ora_f32)), f"max abs diff {np.abs(gpu_f32 - ora_f32).max()}"


def assert_close(gpu, ora, u8=None, ou8=None):
    d = np.abs(gpu - ora) / np.maximum(1.0, np.abs(ora))
    assert d.max() <= TOL, f"max diff {d.max():.3e} at {np.unravel_index(d.argmax(), d.shape)}"
    if u8 is not None:
        assert np.abs(u8.astype(int) - ou8.astype(int)).max() <= 1


@pytest.fixture(scope="module")
def box(gpu):
    return pair(flatten(scenes.box_scene()))


@pytest.fixture(scope="module")
def cornell(gpu):
    return pair(flatten(scenes.cornell_scene()))


def test_box_flat_is_bit_exact_and_matches_golden(box):
    ds, orc = box
    g = np.load(os.path.join(G, "box_256.npz"))
    u8, f32, st = ds.render(FLAT, 256, 256)
    ou8, of32, _ = orc.render(FLAT, 256, 256)
    assert_exact(f32, of32)
    assert_exact(f32, g["flat"])
    assert np.array_equal(u8, ou8)
    assert st["rays_primary"] == 65536 and st["kernel_launches"] == 1


def test_box_pathtracer_depth1_within_tolerance(box):
    ds, orc = box
    g = np.load(os.path.join(G, "box_256.npz"))
    cfg = Config(max_depth=1)
    u8, f32, st = ds.render(cfg, 256, 256, collect_stats=True)
    ou8, of32, ost = orc.render(cfg, 256, 256)
    assert_close(f32, of32, u8, ou8)
    assert_close(f32, g["pt1"])
    assert np.array_equal(f32 == 0, of32 == 0)          # identical hit/miss/shadow decisions
    assert st["rays_shadow"] == ost["rays_shadow"] == 2 * 7744 and st["hits_shaded"] == ost["hits_shaded"]


def test_reference_style_draw_surface(gpu):
    """Reads like rayca-soft/tests/gltf.rs:191-204: SoftRenderer::default().draw(&scene, &mut image)."""
    image = Image(256, 256)
    renderer = SoftRenderer()
    scene = Scene()
    scene.push_model(scenes.load_gltf(os.path.join(G, "box.gltf")))
    scene.push_model(SoftRenderer.create_default_model())
    renderer.draw(scene, image)
    orc = ol.OracleScene(flatten(scene), Config())
    ou8, _, _ = orc.render(Config(), 256, 256)
    assert image.data[..., 3].min() == 255 and (image.data[..., 0] > 0).sum() > 5000
    # default Config = Pathtracer depth 5 with random bounces: same counter-based RNG on both sides;
    # a bounce ray can still split at a silhouette when acos/sin/cos differ in the last bit
    check_u8_outliers("box_default_config_draw_256", image.data, ou8)


def test_triangle_scene_vertex_colour_interpolation(gpu):
    # rayca-soft/tests/gltf.rs:48-84
    ds, orc = pair(flatten(scenes.triangle_scene()))
    _, f32, _ = ds.render(FLAT, 256, 256)
    _, of32, _ = orc.render(FLAT, 256, 256)
    assert_exact(f32, of32)
    assert len(np.unique(f32[..., :3].reshape(-1, 3), axis=0)) > 1000  # a real gradient


@pytest.mark.parametrize("name", ["box", "cornell", "soup1k"])
def test_hit_records_bit_exact(gpu, name):
    if name == "soup1k":
        g = np.load(os.path.join(G, "soup1k_rays.npz"))
        desc = flatten(scenes.soup_scene(1000, extent=0.12))
    elif name == "box":
        g = np.load(os.path.join(G, "box_256.npz"))
        desc = flatten(scenes.box_scene())
    else:
        g = np.load(os.path.join(G, "cornell_128x72.npz"))
        desc = flatten(scenes.cornell_scene())
    ds, orc = pair(desc)
    assert np.array_equal(ds.primitive_order(), orc.primitive_order())
    for trav in (abi.TRAVERSAL_ORDERED, abi.TRAVERSAL_EXHAUSTIVE):
        t, prim, uv, st = ds.trace_rays(g["rays"], traversal=trav, collect_stats=True)
        ot, oprim, ouv, ost = orc.trace_rays(g["rays"])
        assert np.array_equal(prim, oprim) and np.array_equal(bits(t), bits(ot)) and np.array_equal(bits(uv), bits(ouv))
        assert np.array_equal(prim, g["prim"]) and np.array_equal(bits(t), bits(g["t"]))
        if trav == abi.TRAVERSAL_EXHAUSTIVE:
            assert st["triangles_tested"] == ost["triangles_tested"]  # visits exactly the reference's leaves


def test_cornell_1080p_flat_bit_exact(cornell):
    """BASELINE configs[1]: Cornell-box-style scene, primary rays only, 1920x1080."""
    ds, orc = cornell
    u8, f32, st = ds.render(FLAT, 1920, 1080)
    ou8, of32, _ = orc.render(FLAT, 1920, 1080)
    assert_exact(f32, of32)
    assert np.array_equal(u8, ou8)


def test_cornell_depth1_and_bounces(cornell):
    ds, orc = cornell
    cfg = Config(max_depth=1)
    u8, f32, _ = ds.render(cfg, 640, 360)
    ou8, of32, _ = orc.render(cfg, 640, 360)
    assert_close(f32, of32, u8, ou8)
    # 4 bounces (default depth 5): every direction comes from acos/sin/cos, so a handful of paths may
    # take another route; the bulk must still agree to tolerance and the frame means must match
    cfg = Config()
    _, f32, st = ds.render(cfg, 640, 360, collect_stats=True)
    _, of32, ost = orc.render(cfg, 640, 360)
    check_outliers("cornell_640x360_depth5", f32, of32)
    assert abs(float(f32[..., :3].mean()) - float(of32[..., :3].mean())) < 2e-3 * float(of32[..., :3].mean() + 1e-6) + 1e-6
    assert abs(st["rays_bounce"] - ost["rays_bounce"]) <= 0.001 * ost["rays_bounce"]


def test_odd_sizes_and_zero_direction_quirk(box):
    ds, orc = box
    for (w, h) in ((65, 64), (33, 17), (1, 1), (127, 255)):
        _, f32, _ = ds.render(FLAT, w, h)
        _, of32, _ = orc.render(FLAT, w, h)
        assert_exact(f32, of32)
    _, f32, _ = ds.render(FLAT, 65, 64)
    assert np.all(f32[:, 32, :3] == 0)   # centre column: dir.x == 0 -> misses every AABB


def test_samples_per_pixel_and_gamma(box):
    ds, orc = box
    cfg = Config(integrator=IntegratorStrategy.Flat, samples_per_pixel=4, gamma=2.2)
    u8, f32, _ = ds.render(cfg, 128, 128)
    ou8, of32, _ = orc.render(cfg, 128, 128)
    assert_close(f32, of32, u8, ou8)      # powf(1/gamma): device vs host libm
    cfg = Config(integrator=IntegratorStrategy.Flat, samples_per_pixel=4)
    _, f32, _ = ds.render(cfg, 128, 128)
    _, of32, _ = orc.render(cfg, 128, 128)
    assert_exact(f32, of32)               # gamma 1.0: stratified accumulation is exact
    cfg = Config(max_depth=2, samples_per_pixel=2)
    _, f32, _ = ds.render(cfg, 96, 96)
    _, of32, _ = orc.render(cfg, 96, 96)
    check_outliers("box_96_depth2_spp2", f32, of32)


@pytest.mark.parametrize("spp", [2, 3, 5, 9])
def test_subpixel_positions_follow_the_reference_association(box, spp):
    """scene.rs:135-138 evaluates (x + ix*step) + offset.  For spp = 4, 16, ... every term is exact; for 2, 3, 5, 9 the
    association decides the last bit of the pixel coordinate (strate_count = sqrt(spp) is not an integer, or step is not
    a power of two), and with it which side of a silhouette a sample falls on.  Flat frames must stay bit-exact."""
    ds, orc = box
    cfg = Config(integrator=IntegratorStrategy.Flat, samples_per_pixel=spp)
    for (w, h) in ((97, 61), (256, 256)):
        u8, f32, _ = ds.render(cfg, w, h)
        ou8, of32, _ = orc.render(cfg, w, h)
        assert_exact(f32, of32)
        assert np.array_equal(u8, ou8)


def test_bvh_disabled_matches(gpu):
    desc = flatten(scenes.cornell_scene())
    cfg = Config(integrator=IntegratorStrategy.Flat, bvh=False)
    ds, orc = pair(desc, cfg)
    assert ds.info()["max_depth"] <= 1
    _, f32, _ = ds.render(cfg, 320, 180)
    _, of32, _ = orc.render(cfg, 320, 180)
    assert_exact(f32, of32)


def test_large_leaf_chains_soup(gpu):
    """The reference's SAH (boxes seeded at the origin) leaves 100+ triangle leaves: exercises the
    64-primitive leaf chunks and the ordered/exhaustive equivalence."""
    desc = flatten(scenes.soup_scene(20000))
    ds, orc = pair(desc, build=ol.BUILD_BINNED)
    assert np.array_equal(ds.primitive_order(), orc.primitive_order())
    _, a, _ = ds.render(FLAT, 512, 512)
    _, b, _ = ds.render(FLAT, 512, 512, traversal=abi.TRAVERSAL_EXHAUSTIVE)
    _, o, _ = orc.render(FLAT, 512, 512)
    assert_exact(a, o)
    assert_exact(b, o)


def _textured_quad_scene():
    model = Model()
    rs = np.random.RandomState(5)
    tex = rs.randint(0, 256, (8, 8, 4)).astype(np.uint8)
    tex[..., 3] = 255
    img = model.images.push(Image(8, 8, abi.COLOR_RGBA8, tex))
    t = model.textures.push(Texture(image=img))
    mat = model.materials.push(PbrMaterial(color=(0.9, 0.8, 0.7, 1.0), albedo=t, roughness_factor=0.8))
    g = model.geometries.push(TriangleMesh.quad(uv_scale=(3.0, 2.0)))
    p = model.primitives.push(Primitive(geometry=g, material=mat))
    n = model.nodes.push(Node(mesh=model.meshes.push(Mesh(primitives=[p])), trs=Trs(scale=(3.0, 3.0, 1.0))))
    model.root.children.append(n)
    scene = Scene()
    scene.push_model(model)
    scene.push_model(SoftRenderer.create_default_model())
    return scene


def test_albedo_texture_nearest_wrap(gpu):
    # Sampler::sample (rayca-model/src/sampler.rs:11-30) through PbrMaterial::get_color (pbr.rs:94-102)
    ds, orc = pair(flatten(_textured_quad_scene()))
    _, f32, _ = ds.render(FLAT, 200, 200)
    _, of32, _ = orc.render(FLAT, 200, 200)
    assert_exact(f32, of32)
    assert len(np.unique((f32[..., :3] * 255).astype(int).reshape(-1, 3), axis=0)) > 30
    cfg = Config(max_depth=1)
    u8, f32, _ = ds.render(cfg, 200, 200)
    ou8, of32, _ = orc.render(cfg, 200, 200)
    assert_close(f32, of32, u8, ou8)


def _quad_light_room():
    """Phong room lit by an emissive quad light (the SDTF-style setup, light/quad.rs + nee.rs:72-125)."""
    model = Model()
    wall = model.materials.push(PhongMaterial(diffuse=(0.7, 0.7, 0.7, 1.0), specular=(0.1, 0.1, 0.1, 1.0), shininess=8.0))
    emit = model.materials.push(PhongMaterial(emission=(1.0, 1.0, 1.0, 1.0)))
    room = scenes._MeshBuilder()
    X, Y, Z = np.array([2, 0, 0], np.float32), np.array([0, 2, 0], np.float32), np.array([0, 0, 2], np.float32)
    room.grid((-1, 0, -1), Z, X, 1, 1)
    room.grid((-1, 0, -1), X, Y, 1, 1)
    room.grid((-1, 0, -1), Y, Z, 1, 1)
    room.grid((1, 0, -1), Z, Y, 1, 1)
    room.box((-0.4, 0, -0.4), (0.2, 0.7, 0.2))
    g = model.geometries.push(room.mesh())
    p = model.primitives.push(Primitive(geometry=g, material=wall))
    model.root.children.append(model.nodes.push(Node(mesh=model.meshes.push(Mesh(primitives=[p])))))
    lt = model.lights.push(Light.quad(ab=(0.6, 0.0, 0.0), ac=(0.0, 0.0, 0.6), color=(1, 1, 1, 1), material=emit, intensity=6.0))
    # ab x ac = -y: the light faces down
    model.root.children.append(model.nodes.push(Node(light=lt, trs=Trs(translation=(-0.3, 1.9, -0.3)))))
    cam = model.cameras.push(scenes.Camera())
    model.root.children.append(model.nodes.push(Node(camera=cam, trs=Trs(translation=(0.0, 1.0, 3.2)))))
    scene = Scene()
    scene.push_model(model)
    return scene


def test_quad_light_nee_with_phong_materials(gpu):
    desc = flatten(_quad_light_room())
    ds, orc = pair(desc)
    cfg = Config(max_depth=1, light_samples=4, light_stratify=True, seed=3)
    u8, f32, st = ds.render(cfg, 160, 120, collect_stats=True)
    ou8, of32, ost = orc.render(cfg, 160, 120)
    assert st["rays_shadow"] == ost["rays_shadow"] > 0
    check_outliers("quad_light_room_phong_160x120_nee4", f32, of32)   # powf(x, shininess) in the Phong lobe
    assert float(f32[..., :3].max()) > 0.05                    # the light actually reaches the floor
    _, f32, _ = ds.render(FLAT, 160, 120)
    _, of32, _ = orc.render(FLAT, 160, 120)
    assert_exact(f32, of32)                                    # emissive quad visible to primary rays


def test_errors_mirror_the_reference_panics(gpu):
    scene = Scene()
    scene.push_model(scenes.load_gltf(os.path.join(G, "box.gltf")))
    ds = DeviceScene(flatten(scene), Config())
    with pytest.raises(RaycaError) as e:
        ds.render(Config(), 8, 8)
    assert e.value.code == abi.ERR_NO_CAMERA            # assert!(!camera_draw_infos.is_empty())  scene.rs:109
    scene = Scene()
    scene.push_model(SoftRenderer.create_default_model())
    ds = DeviceScene(flatten(scene), Config())
    with pytest.raises(RaycaError) as e:
        ds.render(Config(), 8, 8)
    assert e.value.code == abi.ERR_EMPTY_SCENE          # Tlas::intersects assert  tlas.rs:272
    ds = DeviceScene(flatten(scenes.box_scene()), Config())
    for bad in (Config(direct_sampler=SamplerStrategy.Mis), Config(indirect_sampler=SamplerStrategy.Nee, direct_sampler=SamplerStrategy.NONE)):
        with pytest.raises(RaycaError) as e:
            ds.render(bad, 8, 8)
        assert e.value.code == abi.ERR_UNSUPPORTED       # todo!()/panic arms of the reference: fails loudly, no CPU fallback
    with pytest.raises(RaycaError) as e:
        ds.render(Config(integrator=17), 8, 8)
    assert e.value.code == abi.ERR_BAD_ARG
    with pytest.raises(RaycaError) as e:
        ds.render(Config(), 0, 8)
    assert e.value.code == abi.ERR_BAD_ARG


def test_tiles_reassemble_to_the_full_frame(cornell):
    ds, _ = cornell
    cfg = Config(max_depth=1)
    h, w = 271, 320   # ragged: not a multiple of the band
    _, full, _ = ds.render(cfg, w, h)
    for parts, band in ((2, 8), (3, 16), (8, 8)):
        frame = np.zeros_like(full)
        for part in range(parts):
            _, tile, st = ds.render(cfg, w, h, tile=(part, parts, band))
            rows = [y for y in range(h) if (y // band) % parts == part]
            assert st["rows_rendered"] == len(rows) == tile.shape[0]
            frame[rows] = tile
        assert_exact(frame, full)


# ---- RAYCA_BUILDER_SAH: a different tree, the reference's candidates and tie order --------------------
@pytest.mark.parametrize("name", ["box", "cornell", "soup", "atrium"])
def test_sah_builder_is_exact_against_the_oracle(gpu, name):
    """The SAH tree (empty-seeded candidate boxes) must give the reference's pixels bit for bit: a hit is
    accepted only if the reference's own leaf box passes the slab test, depth ties go to the reference's
    primitive order."""
    desc = {"box": lambda: flatten(scenes.box_scene()), "cornell": lambda: flatten(scenes.cornell_scene()),
            "soup": lambda: flatten(scenes.soup_scene(20000, extent=0.03)), "atrium": lambda: flatten(scenes.atrium_scene(detail=3))}[name]()
    ds = DeviceScene(desc, Config(), builder=abi.BUILDER_SAH)
    orc = ol.OracleScene(desc, Config(), build=ol.BUILD_BINNED)
    w, h = (320, 180) if name != "soup" else (384, 384)
    for trav in (abi.TRAVERSAL_ORDERED, abi.TRAVERSAL_EXHAUSTIVE):
        _, f32, _ = ds.render(FLAT, w, h, traversal=trav)
        _, of32, _ = orc.render(FLAT, w, h)
        assert_exact(f32, of32)
    cfg = Config(max_depth=1)
    u8, f32, _ = ds.render(cfg, w, h)
    ou8, of32, _ = orc.render(cfg, w, h)
    ok = ~(np.isnan(f32) | np.isnan(of32))
    assert np.array_equal(np.isnan(f32), np.isnan(of32))
    assert np.abs(np.where(ok, f32 - of32, 0)).max() <= TOL
    # hit records: same primitive (compared in flatten order), same t and barycentrics
    g = np.load(os.path.join(G, "box_256.npz" if name == "box" else "cornell_128x72.npz"))
    rays = g["rays"] if name in ("box", "cornell") else None
    if rays is not None:
        t, prim, uv, _ = ds.trace_rays(rays)
        ot, oprim, ouv, _ = orc.trace_rays(rays)
        order, oorder = ds.primitive_order(), orc.primitive_order()
        hit = prim != abi.NONE
        assert np.array_equal(hit, oprim != abi.NONE)
        assert np.array_equal(order[prim[hit]], oorder[oprim[hit]])
        assert np.array_equal(bits(t), bits(ot)) and np.array_equal(bits(uv), bits(ouv))


def test_sah_and_reference_builders_agree_at_full_size(gpu):
    """BASELINE configs[2] geometry at 1920x1080: the two trees must produce the same frame (GPU only;
    the oracle would need minutes per frame on the reference's tree)."""
    desc = flatten(scenes.atrium_scene())
    a = DeviceScene(desc, Config(), builder=abi.BUILDER_SAH)
    b = DeviceScene(desc, Config(), builder=abi.BUILDER_REFERENCE)
    assert a.info()["node_count"] > 20 * b.info()["node_count"]   # the reference's tree barely splits (and the SAH layout folds pairs of triangles into one leaf)
    _, fa, _ = a.render(FLAT, 1920, 1080)
    _, fb, _ = b.render(FLAT, 1920, 1080)
    assert_exact(fa, fb)
    _, fa, _ = a.render(FLAT, 1920, 1080, traversal=abi.TRAVERSAL_EXHAUSTIVE)
    assert_exact(fa, fb)
    cfg = Config(max_depth=1)
    ua, fa, sa = a.render(cfg, 1920, 1080, collect_stats=True)
    ub, fb, sb = b.render(cfg, 1920, 1080, collect_stats=True)
    assert np.array_equal(bits(fa), bits(fb)) and np.array_equal(ua, ub)
    assert sa["rays_shadow"] == sb["rays_shadow"] and sa["hits_shaded"] == sb["hits_shaded"]


# ---- spheres (rayca-geometry/src/sphere.rs) -----------------------------------------------------------
def _sphere_scene(with_boxes=False):
    """rayca-soft/tests/gltf.rs:9-46 `sphere` (unit sphere, scale (1,2,1), at z=-1) and, with_boxes, the
    sphere-over-cubes arrangement of `cube_over_plane` (gltf.rs:86-186)."""
    from rayca_amd import Sphere
    scene = Scene()
    model = Model()
    g = model.geometries.push(Sphere.unit())
    mat = model.materials.push(PbrMaterial(color=(0.2, 0.6, 0.9, 1.0), roughness_factor=0.7))
    p = model.primitives.push(Primitive(geometry=g, material=mat))
    m = model.meshes.push(Mesh(primitives=[p]))
    trs = Trs(translation=(-0.5, 2.0, -3.0)) if with_boxes else Trs(translation=(0.0, 0.0, -1.0), scale=(1.0, 2.0, 1.0))
    model.root.children.append(model.nodes.push(Node(mesh=m, trs=trs)))
    scene.push_model(model)
    if with_boxes:
        for shift in ((1.0, 1.0, -2.0), (0.0, 0.0, -1.0), (-1.5, 0.0, -4.0)):
            n = scene.push_model(scenes.load_gltf(os.path.join(G, "box.gltf")))
            scene.nodes[n].trs = Trs(translation=shift)
        floor = scene.push_model(scenes.load_gltf(os.path.join(G, "box.gltf")))
        scene.nodes[floor].trs = Trs(translation=(0.0, -1.0, 0.0), scale=(16.0, 0.125, 16.0))
    scene.push_model(SoftRenderer.create_default_model())
    return scene


@pytest.mark.parametrize("with_boxes", [False, True])
def test_spheres(gpu, with_boxes):
    desc = flatten(_sphere_scene(with_boxes))
    for builder in (abi.BUILDER_REFERENCE, abi.BUILDER_SAH):
        ds = DeviceScene(desc, Config(), builder=builder)
        orc = ol.OracleScene(desc, Config())
        assert ds.info()["sphere_count"] == 1
        _, f32, _ = ds.render(FLAT, 256, 256)
        _, of32, _ = orc.render(FLAT, 256, 256)
        assert_exact(f32, of32)
        assert (f32[..., 2] > 0.8).sum() > 500   # the sphere is there
        cfg = Config(max_depth=1)
        u8, f32, st = ds.render(cfg, 256, 256, collect_stats=True)
        ou8, of32, ost = orc.render(cfg, 256, 256)
        assert_close(f32, of32, u8, ou8)
        assert st["rays_shadow"] == ost["rays_shadow"]
        if builder == abi.BUILDER_SAH:
            # four generations: the split engine, whose bounce rays and bounce shadow rays run on the lane-refill kernels
            # (refill.hip, sphere instantiations) -- against the oracle, and against the fused kernels bit for bit
            cfg4 = Config(max_depth=4)
            _, f4, st4 = ds.render(cfg4, 192, 192, collect_stats=True)
            _, of4, ost4 = orc.render(cfg4, 192, 192)
            check_outliers(f"spheres_depth4_{'boxes' if with_boxes else 'alone'}", f4, of4)
            assert st4["rays_shadow"] == ost4["rays_shadow"] and st4["rays_bounce"] == ost4["rays_bounce"]
            _, ff, _ = ds.render(cfg4, 192, 192, engine=abi.ENGINE_FUSED)
            assert_exact(f4, ff)
        if builder == abi.BUILDER_REFERENCE:
            rs = np.random.RandomState(3)
            o = rs.uniform(-4, 4, (256, 3)).astype(np.float32)
            d = (rs.uniform(-1, 1, (256, 3)) - o * 0.3).astype(np.float32)
            rays = np.concatenate([o, d], 1)
            t, prim, uv, _ = ds.trace_rays(rays)
            ot, oprim, ouv, _ = orc.trace_rays(rays)
            assert np.array_equal(prim, oprim) and np.array_equal(bits(t), bits(ot))


def test_config3_soup_4096_full_size(gpu):
    """BASELINE configs[3] at full size: 1 M random triangles at 4096x4096, primary rays.  Out of the oracle's reach,
    so checked through properties: culled front-to-back traversal == traversal of every leaf the reference visits;
    the first frames of a scene cycle through the four node formats (binary / 4-wide, f32 / fp16) and the two camera-ray
    kernels and must all be the same frame; eight row tiles reassemble it."""
    w = h = 4096
    ds = DeviceScene(flatten(scenes.soup_scene()), Config(), builder=abi.BUILDER_SAH)
    # scene_create leaves the 4-wide / fp16 formats to a thread of its own: a frame issued before they are there runs on
    # the binary f32 nodes (bit 11) and is the frame every later one must equal
    first, _, st = ds.render(FLAT, w, h, want_f32=False, collect_stats=True)   # (a counting frame is never a calibration frame)
    if st["node_format"] & 2048:
        assert not st["node_format"] & (5 | 256 | 512)
    print(f"[soup 4096^2] first frame issued {'before' if st['node_format'] & 2048 else 'after'} the other node formats were there")
    ds.finish()
    # the next frames time the two camera-ray kernels -- fused generation kernel / lane refill (bit 9) -- and then the four
    # node formats in turn on the kernel that won (bit 8): every one of them must be the same frame
    formats, kernels, n = set(), set(), 0
    while True:
        u8, _, st = ds.render(FLAT, w, h, want_f32=False)
        assert np.array_equal(u8, first), (n, st["node_format"])
        if st["node_format"] & 512:
            assert not formats, "a kernel calibration frame after the format calibration had begun"
            kernels.add(bool(st["node_format"] & 1024))
        elif st["node_format"] & 256:
            formats.add(st["node_format"] & 5)
        else:
            break
        n += 1
        assert n < 100, "calibration does not end"
    assert formats == {0, 1, 4, 5} and kernels == {False, True}, (formats, kernels)
    u8, _, st = ds.render(FLAT, w, h, want_f32=False, collect_stats=True)
    assert not st["node_format"] & (256 | 512) and np.array_equal(u8, first)
    print(f"[soup 4096^2] chosen: format bits {st['node_format'] & 5}, lane refill {bool(st['node_format'] & 1024)}, "
          f"node-loop lane utilisation {st['boxes_tested'] / st['wave_box_slots']:.3f}, leaf loop {st['triangles_tested'] / st['wave_triangle_slots']:.3f}")
    ex, _, _ = ds.render(FLAT, w, h, want_f32=False, traversal=abi.TRAVERSAL_EXHAUSTIVE)
    assert np.array_equal(ex, first)
    hit = (first[..., :3].astype(np.int32).sum(-1) > 0).mean()
    assert 0.05 < hit <= 1.0, hit
    frame = np.zeros_like(first)
    for part in range(8):
        rows = [y for y in range(h) if (y // 8) % 8 == part]
        u8, _, sp = ds.render(FLAT, w, h, tile=(part, 8, 8), want_f32=False)
        frame[rows] = u8
    assert np.array_equal(frame, first)
    ds.close()


# ---- GPU BLAS builder (rayca_amd/csrc/bvh_build.hip) ---------------------------------------------------------------
@pytest.mark.parametrize("name", ["atrium", "soup64k", "soup_flat", "soup1m"])
def test_gpu_builder_builds_the_host_builders_tree(gpu, name):
    """The level-by-level GPU build (binned SAH + solved swap partition) against the recursive host build: same
    primitive order, same node count, and -- for rays through the scene -- the same hits AND the same number of box
    and triangle tests, i.e. the same tree, for both seeds of the candidate boxes."""
    if name == "atrium":
        scene = scenes.atrium_scene()
    elif name == "soup64k":
        scene = scenes.soup_scene(1 << 16, extent=0.04)
    elif name == "soup_flat":   # degenerate extent on one axis + many equal centroids: leaves by failed partitions
        scene = scenes.soup_scene(1 << 14, extent=0.0)
    else:
        scene = scenes.soup_scene()
    desc = flatten(scene)
    rs = np.random.RandomState(11)
    o = rs.uniform(-1.5, 1.5, (20000, 3)).astype(np.float32) + np.array([0, 1.0 if name == "atrium" else 0.0, 0], np.float32)
    d = (rs.uniform(-1, 1, (20000, 3))).astype(np.float32)
    rays = np.concatenate([o, d], 1)
    for builder in (abi.BUILDER_REFERENCE, abi.BUILDER_SAH):
        if name == "soup1m" and builder == abi.BUILDER_REFERENCE:
            continue   # the SAH scene builds the reference tree too (ranks, leaves): once is enough at this size
        a = DeviceScene(desc, Config(), builder=builder)
        b = DeviceScene(desc, Config(), builder=builder, build_on_host=True)
        ia, ib = a.info(), b.info()
        assert ia["node_count"] == ib["node_count"] and ia["blas_count"] == ib["blas_count"]
        assert np.array_equal(a.primitive_order(), b.primitive_order())
        ta, pa, ua, sa = a.trace_rays(rays, collect_stats=True)
        tb, pb, ub, sb = b.trace_rays(rays, collect_stats=True)
        assert np.array_equal(pa, pb) and np.array_equal(bits(ta), bits(tb)) and np.array_equal(bits(ua), bits(ub))
        assert sa["boxes_tested"] == sb["boxes_tested"] and sa["triangles_tested"] == sb["triangles_tested"]
        assert name == "soup_flat" or (pa != 0xFFFFFFFF).sum() > 100   # (zero-area triangles are never hit)
        _, fa, _ = a.render(FLAT, 320, 180)
        _, fb, _ = b.render(FLAT, 320, 180)
        assert_exact(fa, fb)
        a.close()
        b.close()


def test_gpu_builder_with_several_large_models(gpu):
    """Two BLASes above the GPU-builder threshold under one TLAS (plus the default model): device build == host build,
    and both match the oracle bit for bit on a Flat frame."""
    scene = scenes.soup_scene(6000, seed=0xA11CE, extent=0.08)
    other = scenes.soup_scene(5000, seed=0xB0B, extent=0.06)
    n = scene.push_model(other.models[0])
    scene.nodes[n].trs = Trs(translation=(0.4, -0.2, 0.3), scale=(0.7, 0.7, 0.7))
    desc = flatten(scene)
    orc = ol.OracleScene(desc, Config())
    _, of32, _ = orc.render(FLAT, 160, 120)
    for builder in (abi.BUILDER_REFERENCE, abi.BUILDER_SAH):
        a = DeviceScene(desc, Config(), builder=builder)
        b = DeviceScene(desc, Config(), builder=builder, build_on_host=True)
        assert a.info()["blas_count"] == 2 and a.info()["node_count"] == b.info()["node_count"]
        assert np.array_equal(a.primitive_order(), b.primitive_order())
        _, fa, _ = a.render(FLAT, 160, 120)
        _, fb, _ = b.render(FLAT, 160, 120)
        assert_exact(fa, fb)
        assert_exact(fa, of32)
        assert (fa[..., :3].sum(-1) > 0).sum() > 500
