"""Generates tests/golden/jpeg/*.jpg and their libjpeg decodes (tests/golden/jpeg/expected.npz) with Pillow (libjpeg-turbo):
baseline and progressive, 4:4:4 / 4:2:2 / 4:2:0, restart intervals, odd sizes.  Run once here; the fixtures are committed, so
tests/test_jpeg.py needs no image library."""
import io, os
import numpy as np
from PIL import Image, features
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "tests", "golden", "jpeg")
os.makedirs(out, exist_ok=True)
rs = np.random.RandomState(7)


def picture(w, h):
    """smooth gradients + a few hard edges + a little noise: every coefficient band gets used"""
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    r = 127 + 120 * np.sin(x / 5.0) * np.cos(y / 7.0)
    g = 255 * (x / max(w - 1, 1))
    b = 255 * (((x // 6) + (y // 5)) % 2)
    im = np.stack([r, g, b], -1) + rs.normal(0, 6, (h, w, 3))
    im[h // 3: h // 3 + 3, :, :] = (250, 10, 10)
    return np.clip(im, 0, 255).astype(np.uint8)


cases = {
    "base_444_q90": dict(size=(40, 32), subsampling=0, quality=90),
    "base_420_odd": dict(size=(37, 29), subsampling=2, quality=85),
    "base_422_odd": dict(size=(35, 18), subsampling=1, quality=75),
    "base_420_restart": dict(size=(64, 48), subsampling=2, quality=80, restart_marker_blocks=3),
    "base_420_optimized": dict(size=(50, 41), subsampling=2, quality=60, optimize=True),
    "prog_420": dict(size=(57, 43), subsampling=2, quality=85, progressive=True),
    "prog_444_restart": dict(size=(33, 40), subsampling=0, quality=92, progressive=True, restart_marker_rows=1),
    "prog_422_q50": dict(size=(48, 31), subsampling=1, quality=50, progressive=True),
    "tiny_420": dict(size=(3, 2), subsampling=2, quality=90),
    "grey": dict(size=(24, 17), quality=90, grey=True),
}
expected = {}
for name, kw in cases.items():
    w, h = kw.pop("size")
    grey = kw.pop("grey", False)
    im = Image.fromarray(picture(w, h))
    if grey:
        im = im.convert("L")
    path = os.path.join(out, name + ".jpg")
    im.save(path, "JPEG", **kw)
    dec = np.asarray(Image.open(path))   # libjpeg-turbo defaults: islow IDCT, fancy upsampling
    expected[name] = dec
    print(name, os.path.getsize(path), "bytes", dec.shape)
np.savez_compressed(os.path.join(out, "expected.npz"), **expected)
print("Pillow", Image.__version__, "libjpeg(-turbo)", features.version("jpg"))
