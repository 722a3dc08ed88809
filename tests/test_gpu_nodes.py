"""The 48-B centre / half-extent node records the conservative kernels traverse (RaycaStats.node_format bit 12,
rayca_hip_scene_read_nodes): every record's boxes CONTAIN the min / max boxes they are made from -- which is all the steering
boxes have to do, exactness comes from the reference-leaf filter -- they are no looser than the 8 mantissa bits their x / y
half extents keep, and the child references stored below those half extents decode to the 64-B node's."""
import math

import numpy as np
import pytest

from rayca_amd import Config, DeviceScene, IntegratorStrategy, abi, flatten, scenes
from rayca_amd.model import PbrMaterial, TriangleMesh, Trs


def far_soup(n=20000, scale=3.0e4, shift=(7.0e5, -2.5e5, 1.0e6)):
    """a soup far from the origin and large: coordinates of 10^5..10^6 with 10^2-sized triangles next to 10^-1-sized ones"""
    rs = np.random.RandomState(3)
    centre = rs.uniform(-1, 1, (n, 1, 3))
    size = np.where(rs.uniform(size=(n, 1, 1)) < 0.5, 3e-3, 3e-6)
    pos = ((centre + rs.uniform(-1, 1, (n, 3, 3)) * size) * scale + np.array(shift)).astype(np.float32).reshape(-1, 3)
    tm = TriangleMesh(pos, np.arange(3 * n, dtype=np.uint32))
    cam = Trs(translation=(shift[0], shift[1], shift[2] + 3.5 * scale))
    scene = scenes._single_model_scene([(tm, PbrMaterial(color=(1, 1, 1, 1), roughness_factor=1.0))], cam, math.pi / 4, [])
    scene.centroids = pos.reshape(-1, 3, 3).astype(np.float64).mean(1)
    return scene


SCENES = {"cornell": scenes.cornell_scene, "atrium": lambda: scenes.atrium_scene(detail=6), "soup64k": lambda: scenes.soup_scene(1 << 16, extent=0.04),
          "soup_flat": lambda: scenes.soup_scene(1 << 14, extent=0.0), "far_soup": far_soup}


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(SCENES))
def test_centre_half_records_contain_their_boxes(gpu, name):
    ds = DeviceScene(flatten(SCENES[name]()), Config(), builder=abi.BUILDER_SAH)
    st = ds.render(Config(integrator=IntegratorStrategy.Flat), 64, 64, want_f32=False, collect_stats=True)[2]
    if not st["node_format"] & 4096:
        pytest.skip("this build traverses the 64-B min / max nodes")
    n64, n48 = ds.read_nodes(0), ds.read_nodes(1)
    assert n64.shape[0] == n48.shape[0] == ds.info()["node_count"] > 0
    L = np.longdouble
    box = n64[:, :12].view(np.float32).astype(L).reshape(-1, 2, 2, 3)       # node, child, (min, max), axis
    rec = n48.view(np.float32).astype(L).reshape(-1, 2, 2, 3)               # node, child, (centre, half), axis
    lo, hi, c, h = box[:, :, 0], box[:, :, 1], rec[:, :, 0], rec[:, :, 1]
    assert np.all(np.isfinite(rec.astype(np.float64)))
    assert np.all(h >= 0)
    assert np.all(c - h <= lo) and np.all(c + h >= hi)
    # no looser than promised: the exact half extent + the centre's rounding, rounded up to 8 mantissa bits with the reference's
    # 16 bits below them (x, y: < 2^-6 of the half extent together) / to f32 (z);
    # below a reference an empty box's zero half extent is a denormal
    exact = (hi - lo) / 2 + np.abs(c - (lo + hi) / 2)
    assert np.all(h <= exact * L(1 + 2.0 ** -6) * L(1 + 1e-6) + L(1e-37))
    assert np.all(h[:, :, 2] <= exact[:, :, 2] * L(1 + 2.0 ** -22) + L(1e-44))
    hb = n48.reshape(-1, 2, 2, 3)[:, :, 1]                                   # the half extents' bits
    ref = (hb[:, :, 0] & 0xFFFF) | ((hb[:, :, 1] & 0xFFFF) << 16)
    want = n64[:, 12:14].copy()
    inner = ((want & 0x80000000) == 0) & (want != 0x7FFFFFFF)
    want[inner] *= 48                                                         # (an inner reference is the child record's byte offset)
    assert np.array_equal(ref, want)
    with pytest.raises(Exception):
        ds.read_nodes(2)
    ds.close()


@pytest.mark.gpu
def test_far_and_large_coordinates_ordered_equals_exhaustive(gpu):
    """the same scene through the kernels: front-to-back traversal on the conservative records == the exhaustive one that
    visits every leaf the reference visits (rays from all around, hits and misses)"""
    scene = far_soup()
    ds = DeviceScene(flatten(scene), Config(), builder=abi.BUILDER_SAH)
    rs = np.random.RandomState(17)
    o = rs.uniform(-1.3, 1.3, (30000, 3)) * 3.0e4 + np.array([7.0e5, -2.5e5, 1.0e6])
    d = rs.normal(size=(30000, 3))
    aim = scene.centroids[rs.randint(0, len(scene.centroids), 20000)]          # two thirds of them at a triangle each
    d[:20000] = aim - o[:20000]
    rays = np.concatenate([o, d / np.linalg.norm(d, axis=1, keepdims=True)], 1).astype(np.float32)
    ta, pa, ua, _ = ds.trace_rays(rays)
    tb, pb, ub, _ = ds.trace_rays(rays, traversal=abi.TRAVERSAL_EXHAUSTIVE)
    assert np.array_equal(pa, pb) and np.array_equal(ta.view(np.uint32), tb.view(np.uint32)) and np.array_equal(ua.view(np.uint32), ub.view(np.uint32))
    assert 100 < (pa != 0xFFFFFFFF).sum() < len(pa)
    ds.close()
