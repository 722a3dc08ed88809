"""JPEG textures: include/rayca_jpeg.hpp against libjpeg-turbo decodes of committed fixtures
(tests/make_jpeg_fixtures.py wrote both with Pillow), and a glTF whose albedo texture is a .jpg through the whole path."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
from rayca_amd import Config, IntegratorStrategy, abi, flatten, model as M, scenes
from rayca_amd.gltf import load_gltf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
J = os.path.join(ROOT, "tests", "golden", "jpeg")
EXPECTED = np.load(os.path.join(J, "expected.npz"))
COLOUR = [n for n in EXPECTED.files if n != "grey"]


@pytest.fixture(scope="module")
def host_mirror(product_lib):
    import __graft_entry__ as g
    return g.build_cpp_host()


def cpp_decode(host_mirror, path, out_dir):
    r = subprocess.run([host_mirror, "image", path, str(out_dir)], capture_output=True, text=True)
    if r.returncode != 0:
        return None, r.stderr
    w, h, ct = np.frombuffer((out_dir / "png_head.bin").read_bytes(), np.uint32)
    ch = 3 if ct == abi.COLOR_RGB8 else 4
    return np.frombuffer((out_dir / "png_texels.bin").read_bytes(), np.uint8).reshape(h, w, ch), ""


@pytest.mark.parametrize("name", COLOUR)
def test_decode_matches_libjpeg(host_mirror, tmp_path, name):
    """baseline / progressive, 4:4:4 / 4:2:2 / 4:2:0, restart intervals, optimised Huffman tables, odd sizes: +-1 per channel
    is the bar; the decoder follows libjpeg's arithmetic, so the decodes are in fact identical."""
    got, err = cpp_decode(host_mirror, os.path.join(J, name + ".jpg"), tmp_path)
    assert got is not None, err
    want = EXPECTED[name]
    assert got.shape == want.shape                         # RGB8, like image::ColorType::Rgb8 -> ColorType::RGB8 (image.rs:195)
    diff = np.abs(got.astype(int) - want.astype(int))
    assert diff.max() <= 1
    assert diff.max() == 0, f"{(diff > 0).sum()} samples differ from libjpeg by 1"


def test_greyscale_is_refused_like_the_reference(host_mirror, tmp_path):
    got, err = cpp_decode(host_mirror, os.path.join(J, "grey.jpg"), tmp_path)
    assert got is None and "Unsupported image color type" in err      # from_image_color_type panics on L8 (image.rs:193-200)


def test_truncated_and_foreign_files_fail_cleanly(host_mirror, tmp_path):
    raw = open(os.path.join(J, "prog_420.jpg"), "rb").read()
    for name, blob in (("cut.jpg", raw[: len(raw) // 3]), ("noise.jpg", b"\xff\xd8\xff" + bytes(range(256)) * 3), ("other.bin", b"GIF89a" + b"\0" * 64)):
        p = tmp_path / name
        p.write_bytes(blob)
        r = subprocess.run([host_mirror, "image", str(p), str(tmp_path)], capture_output=True, text=True)
        assert r.returncode in (0, 1), r.stderr                  # an Error (or a partial picture for the cut file), never a crash
        if name != "cut.jpg":
            assert r.returncode == 1


def _decoder(raw):
    """the libjpeg decode of the one image the glTF fixture refers to (this Python mirror has no JPEG library)"""
    assert raw == open(os.path.join(J, "base_420_odd.jpg"), "rb").read()
    return EXPECTED["base_420_odd"]


def jpg_scene():
    scene = M.Scene()
    scene.push_model(load_gltf(os.path.join(J, "quad_jpg.gltf"), image_decoder=_decoder))
    scene.push_model(M.create_default_model())
    return scene


def _raw(ctypes_array, count, ctype):
    return bytes(C.string_at(ctypes_array, count * C.sizeof(ctype))) if count else b""


def test_gltf_with_a_jpg_albedo_flattens_like_the_python_mirror(host_mirror, tmp_path):
    subprocess.run([host_mirror, "describe", "gltf:" + os.path.join(J, "quad_jpg.gltf"), str(tmp_path)], check=True)
    d = flatten(jpg_scene())
    c = d.c

    def blob(n):
        return (tmp_path / f"{n}.bin").read_bytes()

    assert blob("image_bytes") == d.image_bytes.tobytes() == EXPECTED["base_420_odd"].tobytes()
    assert blob("images") == _raw(c.images, c.image_count, abi.RaycaImage)
    assert blob("textures") == _raw(c.textures, c.texture_count, abi.RaycaTexture)
    assert blob("materials") == _raw(c.materials, c.material_count, abi.RaycaMaterial)
    assert blob("nodes") == _raw(c.nodes, c.node_count, abi.RaycaNode)
    assert blob("uvs") == d.uvs.tobytes() and blob("positions") == d.positions.tobytes()


@pytest.mark.gpu
def test_gltf_with_a_jpg_albedo_renders_like_the_oracle(gpu, host_mirror, tmp_path):
    from rayca_amd import DeviceScene
    desc = flatten(jpg_scene())
    flat = Config(integrator=IntegratorStrategy.Flat)
    for builder in (abi.BUILDER_REFERENCE, abi.BUILDER_SAH):
        ds, orc = DeviceScene(desc, Config(), builder=builder), ol.OracleScene(desc, Config())
        u8, f32, _ = ds.render(flat, 256, 256)
        ou8, of32, _ = orc.render(flat, 256, 256)
        assert np.array_equal(f32.view(np.uint32), of32.view(np.uint32)) and np.array_equal(u8, ou8)
        assert len(np.unique(u8.reshape(-1, 4), axis=0)) > 200          # the picture is on the quad
        cfg = Config(max_depth=2, seed=5)
        _, f32, st = ds.render(cfg, 200, 160, collect_stats=True)
        _, of32, ost = orc.render(cfg, 200, 160)
        assert np.abs(f32 - of32).max() <= 1e-4 * max(1.0, float(np.abs(of32).max()))
        assert st["rays_shadow"] == ost["rays_shadow"]
        ds.close()
        orc.close()
    # the C++ host (its own JPEG decode, its own flatten) draws the frame the Python host draws
    subprocess.run([host_mirror, "draw", "gltf:" + os.path.join(J, "quad_jpg.gltf"), "frame"], check=True, cwd=tmp_path)
    cpp = np.frombuffer((tmp_path / "frame.bin").read_bytes(), np.uint8).reshape(256, 256, 4)
    from rayca_amd import Image, SoftRenderer
    image = Image(256, 256)
    SoftRenderer().draw(jpg_scene(), image)
    assert np.array_equal(cpp, image.data)
