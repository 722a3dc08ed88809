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
