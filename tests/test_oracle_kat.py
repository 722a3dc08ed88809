"""The reference's own known-answer unit tests, restated against the oracle.

Each test cites the Rust test it restates.  These are what pins the oracle: the reference ships no
golden images or vectors for the render path."""
import ctypes as C
import math

import numpy as np

import oracle_lib as ol
from rayca_amd import abi

EPS = np.float32(np.finfo(np.float32).eps) * np.float32(8192.0)  # rayca-math/src/lib.rs:33


def _o3(fn, *args):
    out = (C.c_float * 3)()
    fn(*args, out)
    return np.array(out[:], np.float32)


def close3(a, b):
    """Vec3::close (vec3.rs:108-111) through the oracle."""
    return bool(ol.load().oracle_vec3_close(ol.f3(a), ol.f3(b)))


def test_eps_is_two_to_minus_ten():
    assert float(EPS) == 2.0 ** -10


# rayca-math/src/vec3.rs:520-524
def test_vec3_normalize():
    v = _o3(ol.load().oracle_vec3_normalize, ol.f3((2, 0, 0)))
    assert close3(v, (1, 0, 0))


# rayca-math/src/vec3.rs:527-553
def test_vec3_rotate():
    L = ol.load()
    cases = [((1, 0, 0), (0, 1, 0, 0), (-1, 0, 0)), ((1, 0, 0), (0, 0.707, 0, 0.707), (0, 0, -1)),
             ((1, 0, 0), (0, 0, 1, 0), (-1, 0, 0)), ((1, 0, 0), (0, 0, 0.707, 0.707), (0, 1, 0)),
             ((0, 0, 1), (-0.383, 0, 0, 0.924), (0, 0.707, 0.707))]
    for v, q, want in cases:
        got = _o3(L.oracle_vec3_rotate, ol.f3(v), ol.f4(q))
        assert close3(got, want), (v, q, got, want)


# rayca-math/src/vec3.rs:575-581
def test_vec3_reciprocal_zero_safe():
    r = _o3(ol.load().oracle_vec3_reciprocal, ol.f3((2, 4, 0)))
    assert abs(r[0] - 0.5) < 1e-6 and abs(r[1] - 0.25) < 1e-6 and r[2] == 0.0


# rayca-math/src/vec3.rs:584-589
def test_vec3_reflect():
    r = _o3(ol.load().oracle_vec3_reflect, ol.f3((1, -1, 0)), ol.f3((0, 1, 0)))
    assert close3(r, (1, 1, 0))


# rayca-math/src/vec3.rs:592-598
def test_vec3_dot_and_cross():
    L = ol.load()
    assert L.oracle_vec3_dot(ol.f3((1, 0, 0)), ol.f3((0, 1, 0))) == 0.0
    c = _o3(L.oracle_vec3_cross, ol.f3((1, 0, 0)), ol.f3((0, 1, 0)))
    assert close3(c, (0, 0, 1))


def test_vec3_close_is_lexicographic():
    # derived PartialOrd on f32x4 compares lanes lexicographically, so only x decides
    # unless |dx| == EPS exactly.
    assert close3((0, 5, 5), (0, 0, 0))          # |dx| = 0 < EPS -> Less at lane 0
    assert not close3((1, 0, 0), (0, 0, 0))      # |dx| > EPS -> Greater at lane 0
    assert close3((float(EPS), 0.0001, 0), (0, 0, 0))      # lane0 equal, lane1 0.0001 < EPS
    assert not close3((float(EPS), 0.5, 0), (0, 0, 0))     # lane0 equal, lane1 0.5 > EPS


# rayca-math/src/quat.rs tests: Hamilton product, composition order (quat.rs:236-258)
def test_quat_mul_identity_and_composition():
    L = ol.load()
    out = (C.c_float * 4)()
    L.oracle_quat_mul(ol.f4((0, 0, 0, 1)), ol.f4((0.1, 0.2, 0.3, 0.9)), out)
    assert np.allclose(out[:], (0.1, 0.2, 0.3, 0.9))
    s = math.sqrt(0.5)
    L.oracle_quat_mul(ol.f4((0, s, 0, s)), ol.f4((0, s, 0, s)), out)  # 90 deg about Y twice = 180 deg
    assert np.allclose(out[:], (0, 1, 0, 0), atol=1e-6)
    # i * j = k
    L.oracle_quat_mul(ol.f4((1, 0, 0, 0)), ol.f4((0, 1, 0, 0)), out)
    assert np.allclose(out[:], (0, 0, 1, 0))


# rayca-math/src/trs.rs:461-504
def test_trs_compose():
    L = ol.load()
    out = abi.RaycaTrs()
    L.oracle_trs_mul(C.byref(ol.trs()), C.byref(ol.trs()), C.byref(out))
    assert list(out.translation) == [0, 0, 0] and list(out.rotation) == [0, 0, 0, 1] and list(out.scale) == [1, 1, 1]
    L.oracle_trs_mul(C.byref(ol.trs((1, 2, 3))), C.byref(ol.trs((4, 5, 6))), C.byref(out))
    assert list(out.translation) == [5, 7, 9]
    L.oracle_trs_mul(C.byref(ol.trs(scale=(2, 2, 2))), C.byref(ol.trs(scale=(0.5, 0.5, 0.5))), C.byref(out))
    assert list(out.scale) == [1, 1, 1]
    s = math.sqrt(0.5)
    L.oracle_trs_mul(C.byref(ol.trs(rotation=(0, s, 0, s))), C.byref(ol.trs(rotation=(0, s, 0, s))), C.byref(out))
    assert np.allclose(out.rotation[:], (0, 1, 0, 0), atol=1e-5)


# rayca-math/src/trs.rs:506-553
def test_trs_inverse_roundtrip():
    L = ol.load()
    s = math.sqrt(0.5)
    cases = [(ol.trs((1, 2, 3)), (4, 5, 6)), (ol.trs(scale=(2, 3, 4)), (8, 9, 12)), (ol.trs(rotation=(0, 0, s, s)), (1, 0, 0)),
             (ol.trs((1, 2, 3), (0, 0, s, s), (2, 2, 2)), (1, 1, 1))]
    for t, v in cases:
        fwd = _o3(L.oracle_trs_vec, C.byref(t), ol.f3(v))
        back = _o3(L.oracle_inv_trs_vec, C.byref(t), ol.f3(fwd))
        assert close3(back, v), (v, fwd, back)


# rayca-math/src/ray.rs:158-194
def test_ray_rotate_scale_translate():
    L = ol.load()
    out = (C.c_float * 9)()
    L.oracle_trs_ray(C.byref(ol.trs(rotation=(-0.383, 0, 0, 0.924))), ol.f3((0, 0, 0)), ol.f3((0, 0, -1)), out)
    assert close3(out[3:6], (0, -0.707, -0.707))
    L.oracle_trs_ray(C.byref(ol.trs(scale=(2, 2, 2))), ol.f3((0, 0, 0)), ol.f3((0, 0, -1)), out)
    assert close3(out[3:6], (0, 0, -2))
    rd = _o3(L.oracle_vec3_reciprocal, ol.f3(out[3:6]))
    assert np.allclose(out[6:9], rd, atol=1e-5)
    L.oracle_trs_ray(C.byref(ol.trs((1, 2, 3))), ol.f3((0, 0, 0)), ol.f3((0, 0, -1)), out)
    assert close3(out[0:3], (1, 2, 3))


# rayca-math/src/color/rgba8.rs:110-132
def test_rgba8_from_color():
    L = ol.load()
    out = (C.c_uint8 * 4)()
    L.oracle_rgba8_from_color(ol.f4((1.0, 0.0, 0.5, 1.0)), out)
    assert list(out) == [255, 0, 127, 255]
    L.oracle_rgba8_from_color(ol.f4((2.0, -1.0, 0.5, 1.5)), out)
    assert list(out) == [255, 0, 127, 255]
    L.oracle_rgba8_from_color(ol.f4((float("nan"), 0.999, 0.0039, 1.0)), out)
    assert list(out) == [0, 254, 0, 255]  # NaN -> 0; `as u8` truncates


def test_color_add_multiplies_rhs_by_its_alpha():
    # color/mod.rs:239-248
    L = ol.load()
    out = (C.c_float * 4)()
    L.oracle_color_add(ol.f4((0.1, 0.2, 0.3, 1.0)), ol.f4((1.0, 1.0, 1.0, 0.5)), out)
    assert np.allclose(out[:], (0.6, 0.7, 0.8, 1.0))


# rayca-geometry/src/triangle.rs:569-576
def test_triangle_intersect():
    L = ol.load()
    tri = (C.c_float * 9)(-1, 0, 0, 1, 0, 0, 0, 1, 0)
    t = C.c_float()
    uv = (C.c_float * 2)()
    p = (C.c_float * 3)()
    assert L.oracle_triangle_intersects(tri, C.byref(ol.trs()), ol.f3((0, 0, 1)), ol.f3((0, 0, -1)), C.byref(t), uv, p) == 1
    assert t.value == 1.0 and list(p) == [0, 0, 0]
    # barycentrics: u weights vertex 0, v vertex 1 (bvh/triangle.rs:34-38); at (0,0,0): u = v = 0.5
    assert abs(uv[0] - 0.5) < 1e-6 and abs(uv[1] - 0.5) < 1e-6
    assert L.oracle_triangle_intersects(tri, C.byref(ol.trs()), ol.f3((0, 0, 1)), ol.f3((0, 0, 1)), C.byref(t), uv, p) == 0
    # back-face culling is always on (triangle.rs:96-98)
    assert L.oracle_triangle_intersects(tri, C.byref(ol.trs()), ol.f3((0, 0, -1)), ol.f3((0, 0, 1)), C.byref(t), uv, p) == 0


# rayca-geometry/src/sphere.rs:185-201
def test_sphere_intersect():
    L = ol.load()
    t = C.c_float()
    p = (C.c_float * 3)()
    I = ol.trs()
    assert L.oracle_sphere_intersects(ol.f3((0, 0, 0)), 1.0, C.byref(I), ol.f3((0, 0, 0)), ol.f3((1, 0, 0)), C.byref(t), p) == 1
    assert L.oracle_sphere_intersects(ol.f3((0, 0, 0)), 1.0, C.byref(I), ol.f3((2, 0, 0)), ol.f3((1, 0, 0)), C.byref(t), p) == 0
    assert L.oracle_sphere_intersects(ol.f3((4, 0, 0)), 1.0, C.byref(I), ol.f3((0, 0, 0)), ol.f3((0, 0, -1)), C.byref(t), p) == 0


def test_aabb_slab_and_zero_direction_quirk():
    # bvh/aabb.rs:74-93.  SURVEY quirk 1: a ray with an exactly-zero direction component misses
    # every box, because rdir = 0 there gives t1 = t2 = 0 and tmax <= 0.
    L = ol.load()
    FMAX = float(np.finfo(np.float32).max)
    a, b = ol.f3((-1, -1, -1)), ol.f3((1, 1, 1))
    tmin = L.oracle_aabb_intersects(a, b, ol.f3((0.1, 0.2, 5)), ol.f3((0.01, 0.02, -1)))
    assert abs(tmin - 4.0) < 1e-5
    assert L.oracle_aabb_intersects(a, b, ol.f3((0, 0, 5)), ol.f3((0, 0, -1))) == FMAX  # zero components
    assert L.oracle_aabb_intersects(a, b, ol.f3((0.1, 0.2, 5)), ol.f3((0.01, 0.02, 1))) == FMAX  # behind


# rayca-model/src/sampler.rs:36-50
def test_sampler_wrap():
    L = ol.load()
    assert L.oracle_sampler_wrap(0.0, 1) == 0 and L.oracle_sampler_wrap(1.0, 1) == 0
    assert L.oracle_sampler_wrap(0.5, 4) == 2 and L.oracle_sampler_wrap(1.25, 4) == 1 and L.oracle_sampler_wrap(-0.25, 4) == 3


def test_rng_is_a_pure_function_with_23_bit_floats():
    L = ol.load()
    k = L.oracle_rng_root(7, 1234, 0)
    assert k == L.oracle_rng_root(7, 1234, 0) and k != L.oracle_rng_root(7, 1235, 0)
    vals = np.array([L.oracle_rng_f32(k, d) for d in range(2000)], np.float32)
    assert (vals >= 0).all() and (vals < 1).all()
    assert np.all((vals * np.float32(2 ** 23)) % 1 == 0)  # multiples of 2^-23, like fastrand::f32
    assert abs(float(vals.mean()) - 0.5) < 0.03
    assert L.oracle_rng_child(k, 0) != L.oracle_rng_child(k, 1)


# ================================================================================================
# Round 2: the remaining reference unit tests of SURVEY.md section 4, restated assertion by assertion.
# ================================================================================================
def _m16(fn, *args):
    out = (C.c_float * 16)()
    fn(*args, out)
    return np.array(out[:], np.float32)


def _q4(fn, *args):
    out = (C.c_float * 4)()
    fn(*args, out)
    return np.array(out[:], np.float32)


def f16(v):
    return (C.c_float * 16)(*[float(x) for x in np.asarray(v, np.float32).reshape(-1)])


# rayca-geometry/src/triangle.rs:578-593  centroid_and_min_max (Triangle::default = (-1,0,0) (1,0,0) (0,1,0))
def test_triangle_centroid_and_min_max():
    L = ol.load()
    tri = (C.c_float * 9)(-1, 0, 0, 1, 0, 0, 0, 1, 0)
    c, mn, mx = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)()
    L.oracle_triangle_bounds(tri, C.byref(ol.trs()), c, mn, mx)
    f = np.float32
    expected = (np.array([-1, 0, 0], f) + np.array([1, 0, 0], f) + np.array([0, 1, 0], f)) * f(0.3333)
    assert float(np.sqrt(((np.array(c[:], f) - expected) ** 2).sum())) < 1e-3
    # the reference multiplies by 0.3333, not by 1/3: y is exactly f32(0.3333)
    assert c[1] == float(f(0.3333)) and c[1] != float(f(1.0) / f(3.0))
    assert list(mn) == [-1.0, 0.0, 0.0]
    assert list(mx) == [1.0, 1.0, 0.0]


# rayca-geometry/src/sphere.rs:203-213  min_max_and_centroid
def test_sphere_min_max_and_centroid():
    L = ol.load()
    c, mn, mx = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)()
    L.oracle_sphere_bounds(ol.f3((1, 2, 3)), 2.0, C.byref(ol.trs()), c, mn, mx)
    assert list(mn) == [-1.0, 0.0, 1.0]
    assert list(mx) == [3.0, 4.0, 5.0]
    assert list(c) == [1.0, 2.0, 3.0]


# rayca-soft/src/bvh/triangle.rs:83-114  intersect: a unit TriangleMesh reached through Model -> Scene ->
# SceneDrawInfo -> BvhPrimitive::from_mesh, then BvhPrimitive::intersects(&scene_draw_info, &ray) -- no BVH
def test_bvh_triangle_intersect_through_the_scene_graph():
    from rayca_amd import model as M
    L = ol.load()
    model = M.Model()
    geometry = model.geometries.push(M.TriangleMesh.unit())
    prim = model.primitives.push(M.Primitive(geometry=geometry))
    mesh = model.meshes.push(M.Mesh(primitives=[prim]))
    node = model.nodes.push(M.Node(mesh=mesh))
    model.root.children.append(node)
    scene = M.Scene()
    mh = scene.models.push(model)
    nh = scene.nodes.push(M.Node(model=mh))
    scene.root.children.append(nh)
    orc = ol.OracleScene(M.flatten(scene))
    assert orc.primitive_count == 1
    t = C.c_float()
    uv = (C.c_float * 2)()
    assert L.oracle_scene_primitive_intersects(orc.handle, 0, ol.f3((0, 0, 1)), ol.f3((0, 0, -1)), C.byref(t), uv) == 1
    assert t.value == 1.0
    assert L.oracle_scene_primitive_intersects(orc.handle, 0, ol.f3((0, 0, 1)), ol.f3((0, 0, 1)), C.byref(t), uv) == 0
    # the same two rays through Tlas::intersects miss BOTH times: a zero direction component misses every AABB
    #  -- which is why the reference's test goes to the primitive directly
    tt, prim_id, _, _ = orc.trace_rays(np.array([[0, 0, 1, 0, 0, -1], [0, 0, 1, 0, 0, 1]], np.float32))
    assert (prim_id == abi.NONE).all()


# rayca-math/src/mat4.rs:325-339  mul
def test_mat4_mul():
    L = ol.load()
    a = _m16(L.oracle_mat4_identity)
    b = _m16(L.oracle_mat4_from_scale, ol.f3((2, 2, 2)))     # identity().scale(2,2,2)
    assert b[0] == 2.0 and b[5] == 2.0 and b[10] == 2.0
    c = _m16(L.oracle_mat4_mul, f16(a), f16(b))
    assert not np.array_equal(c, a)
    assert c[0] == 2.0 and c[5] == 2.0 and c[10] == 2.0


# rayca-math/src/mat4.rs:341-349  look_at
def test_mat4_look_at():
    L = ol.load()
    eye = (0.0, 0.0, 4.0)
    m = _m16(L.oracle_mat4_look_at, ol.f3((0, 0, 0)), ol.f3(eye), ol.f3((0, 1, 0)))
    translation = np.array([m[3], m[7], m[11]], np.float32)   # Mat4::get_translation  mat4.rs:121-123
    assert list(-translation) == list(eye)
    q = _q4(L.oracle_mat4_get_rotation, f16(m))
    assert list(q) == [0.0, 0.0, 0.0, 1.0]                    # Quat::default()


# rayca-math/src/mat4.rs:351-363  transmute (row-major storage) and :365-378 transpose
def test_mat4_layout_and_transpose():
    L = ol.load()
    rows = np.arange(16, dtype=np.float32).reshape(4, 4)
    assert [float(x) for x in f16(rows)] == [float(i) for i in range(16)]
    cols = np.array([[0, 4, 8, 12], [1, 5, 9, 13], [2, 6, 10, 14], [3, 7, 11, 15]], np.float32)
    t = _m16(L.oracle_mat4_transpose, f16(cols))
    assert list(t) == [float(i) for i in range(16)]


# rayca-math/src/mat4.rs:380-386  identity_is_identity
def test_mat4_identity_times_vec3():
    L = ol.load()
    ident = _m16(L.oracle_mat4_identity)
    assert list(_o3(L.oracle_mat4_mul_vec3, f16(ident), ol.f3((1, 2, 3)))) == [1.0, 2.0, 3.0]


# rayca-math/src/mat4.rs:388-402  scale_and_get_scale, translation_and_get_translation
def test_mat4_scale_and_translation_getters():
    L = ol.load()
    m = _m16(L.oracle_mat4_from_scale, ol.f3((3, 4, 5)))
    assert [m[0], m[5], m[10]] == [3.0, 4.0, 5.0]            # get_scale  mat4.rs:113-115
    m = _m16(L.oracle_mat4_from_translation, ol.f3((7, 8, 9)))
    assert [m[3], m[7], m[11]] == [7.0, 8.0, 9.0]            # get_translation


# rayca-math/src/mat4.rs:404-421  set_and_get (row*4+col addressing), from_and_as_slices
def test_mat4_row_major_addressing():
    L = ol.load()
    arr = np.arange(1, 17, dtype=np.float32).reshape(4, 4)
    ident = _m16(L.oracle_mat4_identity).reshape(4, 4)
    # a matrix survives a multiplication by the identity element for element: rows stay rows
    out = _m16(L.oracle_mat4_mul, f16(ident), f16(arr)).reshape(4, 4)
    assert np.array_equal(out, arr)
    m = ident.copy()
    m[2, 3] = 42.0                                            # set(2, 3, 42.0)
    p = _o3(L.oracle_mat4_mul_point3, f16(m), ol.f3((0, 0, 0)))
    assert list(p) == [0.0, 0.0, 42.0]                        # (2,3) is the z translation slot: get(2,3) == 42


# rayca-math/src/vec3.rs:556-564  arithmetic
def test_vec3_arithmetic():
    L = ol.load()
    outs = [(C.c_float * 3)() for _ in range(5)]
    L.oracle_vec3_arith(ol.f3((1, 2, 3)), ol.f3((4, 5, 6)), 2.0, *outs)
    add, sub, mul, div, neg = [list(o) for o in outs]
    assert add == [5.0, 7.0, 9.0]
    assert sub == [3.0, 3.0, 3.0]
    assert mul == [2.0, 4.0, 6.0]
    assert div == [2.0, 2.5, 3.0]
    assert neg == [-1.0, -2.0, -3.0]


# rayca-math/src/vec3.rs:566-573  min_max
def test_vec3_min_max():
    L = ol.load()
    mn, mx = (C.c_float * 3)(), (C.c_float * 3)()
    L.oracle_vec3_min_max(ol.f3((1, 5, 3)), ol.f3((4, 2, 6)), mn, mx)
    assert list(mn) == [1.0, 2.0, 3.0] and list(mx) == [4.0, 5.0, 6.0]


def _axis_angle(axis, angle):
    return _q4(ol.load().oracle_quat_axis_angle, ol.f3(axis), float(angle))


def _qmul(a, b):
    return _q4(ol.load().oracle_quat_mul, ol.f4(a), ol.f4(b))


# rayca-math/src/quat.rs:301-311  invert
def test_quat_invert():
    L = ol.load()
    f = np.float32
    a = np.array([0.0, np.sin(f(math.pi / 4)), 0.0, np.cos(f(math.pi / 4))], f)
    assert L.oracle_quat_is_normalized(ol.f4(a))              # get_inverse asserts it (quat.rs:101)
    b = _q4(L.oracle_quat_conjugate, ol.f4(a))
    assert a[0] == b[0] and a[1] == -b[1] and a[2] == b[2] and a[3] == b[3]
    assert L.oracle_quat_is_normalized(ol.f4(b))


# rayca-math/src/quat.rs:313-321  identity_and_getters
def test_quat_identity():
    L = ol.load()
    q = (0.0, 0.0, 0.0, 1.0)
    assert L.oracle_quat_is_normalized(ol.f4(q))
    assert list(_qmul(q, q)) == [0.0, 0.0, 0.0, 1.0]


# rayca-math/src/quat.rs:323-329  axis_angle_normalization
def test_quat_axis_angle_is_normalized():
    q = _axis_angle((1, 0, 0), np.float32(math.pi))
    assert ol.load().oracle_quat_is_normalized(ol.f4(q))


# rayca-math/src/quat.rs:331-340  conjugate_and_inverse
def test_quat_conjugate_and_inverse():
    L = ol.load()
    qn = _q4(L.oracle_quat_normalize, ol.f4((1, 2, 3, 4)))
    conj = _q4(L.oracle_quat_conjugate, ol.f4(qn))
    assert L.oracle_quat_is_normalized(ol.f4(qn))             # so get_inverse == get_conjugate (quat.rs:99-103)
    assert np.array_equal(conj, qn * np.array([-1, -1, -1, 1], np.float32))
    assert L.oracle_quat_is_normalized(ol.f4(conj))


# rayca-math/src/quat.rs:342-348  mul_quat_identity
def test_quat_mul_identity_exact():
    q = _axis_angle((0, 1, 0), np.float32(math.pi / 2))
    ident = (0.0, 0.0, 0.0, 1.0)
    assert np.array_equal(_qmul(q, ident), q)
    assert np.array_equal(_qmul(ident, q), q)


# rayca-math/src/quat.rs:350-355  dot_and_len
def test_quat_dot_and_len():
    L = ol.load()
    q = ol.f4((1, 0, 0, 0))
    assert L.oracle_quat_dot(q, q) == 1.0
    assert L.oracle_quat_len(q) == 1.0


# rayca-math/src/quat.rs:357-371  hamilton_product_vs_expected
def test_quat_product_is_not_commutative():
    qx = _axis_angle((1, 0, 0), np.float32(math.pi / 2))
    qy = _axis_angle((0, 1, 0), np.float32(math.pi / 4))
    assert not np.array_equal(_qmul(qx, qy), _qmul(qy, qx))


# rayca-math/src/quat.rs:373-388  hamilton_product_known_values: q2 * (q1 * v) == (q2 * q1) * v
def test_quat_composition_matches_sequential_rotation():
    L = ol.load()
    q1 = _axis_angle((1, 0, 0), np.float32(math.pi / 2))
    q2 = _axis_angle((0, 1, 0), np.float32(math.pi / 2))
    v = (1.0, 0.0, 0.0)
    v_seq = _o3(L.oracle_vec3_rotate, ol.f3(_o3(L.oracle_vec3_rotate, ol.f3(v), ol.f4(q1))), ol.f4(q2))
    v_prod = _o3(L.oracle_vec3_rotate, ol.f3(v), ol.f4(_qmul(q2, q1)))
    assert float(np.abs(v_seq - v_prod).sum()) < 1e-4


# rayca-math/src/quat.rs:390-399  rotate_vector_by_quaternion
def test_quat_rotates_x_to_y_about_z():
    q = _axis_angle((0, 0, 1), np.float32(math.pi / 2))
    r = _o3(ol.load().oracle_vec3_rotate, ol.f3((1, 0, 0)), ol.f4(q))
    assert abs(r[0] - 0.0) < 1e-5 and abs(r[1] - 1.0) < 1e-5 and abs(r[2] - 0.0) < 1e-5
