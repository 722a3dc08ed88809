"""BVH build of the oracle: the reference's structural unit tests restated, and the binned sweep
checked against the literal 63x3-plane evaluation."""
import numpy as np
import pytest

import oracle_lib as ol
from rayca_amd import (Config, Mesh, Model, Node, Primitive, Scene, TriangleMesh, Trs, flatten, scenes)


def _scene_with(meshes):
    model = Model()
    prims = [model.primitives.push(Primitive(geometry=model.geometries.push(m))) for m in meshes]
    node = model.nodes.push(Node(mesh=model.meshes.push(Mesh(primitives=prims))))
    model.root.children.append(node)
    scene = Scene()
    scene.push_model(model)
    return scene


# rayca-soft/src/bvh/blas.rs:403-434 `simple`: one triangle -> root stays a leaf
def test_single_triangle_root_is_leaf():
    o = ol.OracleScene(flatten(_scene_with([TriangleMesh.unit()])), Config())
    assert o.blas_count == 1
    boxes, rng = o.blas_nodes(0)
    assert rng[0, 1] == 1 and rng[0, 0] == 0          # root holds the primitive
    assert len(rng) == 2                               # root + the unused slot 1 (blas.rs:254-256)
    assert np.allclose(boxes[0, :3], (-1, 0, 0)) and np.allclose(boxes[0, 4:7], (1, 1, 0))


# rayca-soft/src/bvh/blas.rs:436-490 `two_children`: two separated triangles -> root splits
def test_two_separated_triangles_split():
    far = TriangleMesh(np.array([[-4, 0, 0], [-2, 0, 0], [-3, 0.3, 0]], np.float32), np.array([0, 1, 2], np.uint8))
    o = ol.OracleScene(flatten(_scene_with([TriangleMesh.unit(), far])), Config())
    boxes, rng = o.blas_nodes(0)
    assert rng[0, 1] == 0 and rng[0, 0] == 2           # inner: count 0, left child index 2
    assert len(rng) == 4 and rng[2, 1] == 1 and rng[3, 1] == 1


def test_bvh_false_keeps_one_leaf_per_model():
    # scene.rs:96-98: !config.bvh -> max_depth(0)
    o = ol.OracleScene(flatten(scenes.cornell_scene()), Config(bvh=False))
    _, rng = o.blas_nodes(0)
    assert len(rng) == 2 and rng[0, 1] == 36


def test_sah_boxes_start_at_the_origin_quirk():
    # blas.rs:66-67 + aabb.rs:9-13: split candidates are priced with boxes that contain the origin,
    # so a cluster far from the origin is never split however many triangles it has.
    rs = np.random.RandomState(1)
    c = rs.uniform(-0.5, 0.5, (200, 1, 3)).astype(np.float32) + np.array([[[50, 50, 50]]], np.float32)
    pos = (c + rs.uniform(-0.05, 0.05, (200, 3, 3)).astype(np.float32)).reshape(-1, 3)
    o = ol.OracleScene(flatten(_scene_with([TriangleMesh(pos, np.arange(600, dtype=np.uint16))])), Config())
    _, rng = o.blas_nodes(0)
    assert rng[0, 1] == 200


@pytest.mark.parametrize("name,n", [("box", 0), ("cornell", 0), ("soup", 3000), ("atrium", 1)])
def test_binned_sweep_equals_literal_sah(name, n):
    if name == "box":
        d = flatten(scenes.box_scene())
    elif name == "cornell":
        d = flatten(scenes.cornell_scene())
    elif name == "soup":
        d = flatten(scenes.soup_scene(n))
    else:
        d = flatten(scenes.atrium_scene(detail=n))
    a = ol.OracleScene(d, Config(), build=ol.BUILD_LITERAL, xform=ol.XFORM_PER_TEST)
    b = ol.OracleScene(d, Config(), build=ol.BUILD_BINNED)
    assert a.blas_count == b.blas_count
    assert np.array_equal(a.primitive_order(), b.primitive_order())
    for i in range(a.blas_count):
        ba, ra = a.blas_nodes(i)
        bb, rb = b.blas_nodes(i)
        assert np.array_equal(ra, rb)
        assert np.array_equal(ba.view(np.uint32), bb.view(np.uint32))  # boxes bit-identical


def test_cached_world_vertices_equal_per_test_transform():
    """Pre-transformed vertices are bit-identical to what Triangle::intersects recomputes per ray
    (the basis for uploading world-space triangles to the GPU)."""
    d = flatten(scenes.box_scene())
    cfg = Config(max_depth=2)
    a = ol.OracleScene(d, cfg, xform=ol.XFORM_PER_TEST)
    b = ol.OracleScene(d, cfg, xform=ol.XFORM_CACHED)
    _, fa, _ = a.render(cfg, 96, 96)
    _, fb, _ = b.render(cfg, 96, 96)
    assert np.array_equal(fa.view(np.uint32), fb.view(np.uint32))
