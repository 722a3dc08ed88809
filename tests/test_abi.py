"""The ctypes mirror must match include/rayca_hip.h byte for byte, and librayca_hip.so must load and
export every declared symbol (no compute calls here: there is no GPU in this container)."""
import ctypes as C
import os
import re
import subprocess
import tempfile

from rayca_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STRUCTS = ["RaycaConfig", "RaycaTrs", "RaycaNode", "RaycaMesh", "RaycaPrimitive", "RaycaMaterial", "RaycaTexture",
           "RaycaImage", "RaycaCamera", "RaycaLight", "RaycaSceneDesc", "RaycaBuildOptions", "RaycaTile",
           "RaycaRenderOptions", "RaycaStats", "RaycaSceneInfo"]


def test_struct_layout_matches_header():
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{ROOT}/include/rayca_hip.h"', "int main(void){"]
    for s in STRUCTS:
        lines.append(f'printf("{s} %zu\\n", sizeof({s}));')
        for name, _ in getattr(abi, s)._fields_:
            lines.append(f'printf("{s}.{name} %zu\\n", offsetof({s}, {name}));')
    lines.append("return 0;}")
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "l.c"), os.path.join(d, "l")
        open(src, "w").write("\n".join(lines))
        subprocess.run(["gcc", "-std=c11", "-o", exe, src], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    want = dict(l.split() for l in out.strip().splitlines())
    for s in STRUCTS:
        cls = getattr(abi, s)
        assert C.sizeof(cls) == int(want[s]), s
        for name, _ in cls._fields_:
            assert getattr(cls, name).offset == int(want[f"{s}.{name}"]), f"{s}.{name}"


def test_header_declares_exactly_the_bound_symbols():
    text = open(os.path.join(ROOT, "include", "rayca_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(rayca_hip_[a-z_0-9]+)\s*\(", text)))
    assert declared == sorted(abi.PRODUCT_SYMBOLS)


def test_library_loads_and_exports_every_symbol(product_lib):
    for sym in abi.PRODUCT_SYMBOLS:
        assert getattr(product_lib, sym) is not None
    assert product_lib.rayca_hip_version() == abi.ABI_VERSION
    cfg = abi.RaycaConfig()
    product_lib.rayca_hip_config_default(C.byref(cfg))
    # Config::default()  rayca-soft/src/config.rs:10-49
    assert (cfg.bvh, cfg.light_samples, cfg.light_stratify, cfg.samples_per_pixel, cfg.russian_roulette) == (1, 1, 0, 1, 0)
    assert (cfg.direct_sampler, cfg.indirect_sampler, cfg.integrator, cfg.max_depth, cfg.gamma) == (
        abi.SAMPLER_NEE, abi.SAMPLER_COSINE, abi.INTEGRATOR_PATHTRACER, 5, 1.0)


def test_tile_rows_helper(product_lib):
    t = abi.RaycaTile()
    for height in (1, 7, 64, 1080, 2160):
        for parts in (1, 2, 3, 8):
            total = 0
            for part in range(parts):
                t.part, t.parts, t.band_rows = part, parts, 8
                total += product_lib.rayca_hip_tile_rows(C.byref(t), height)
            assert total == height


def test_no_device_fails_loudly_not_silently(product_lib):
    """Without a GPU scene creation must return an error (never a CPU result)."""
    if product_lib.rayca_hip_device_count() > 0:
        return
    from rayca_amd import flatten, scenes
    from rayca_amd.lib import last_error
    d = flatten(scenes.triangle_scene())
    h = C.c_void_p()
    rc = product_lib.rayca_hip_scene_create(d.ptr(), None, None, C.byref(h))
    assert rc == abi.ERR_NO_DEVICE and not h.value
    assert "no CPU fallback" in last_error()


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under rayca_amd/ or in the C ABI sources may
    reference it."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "rayca_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".inc", ".h")):
                txt = open(os.path.join(base, f), errors="replace").read()
                if re.search(r"oracle", txt, re.I):
                    bad.append(os.path.join(base, f))
    assert not bad, bad


def test_host_selftest_fp16_outward_rounding(product_lib):
    """The fp16 steering boxes must contain their f32 originals: the directed rounding is checked exhaustively over
    all finite halves inside the library (host code, no GPU)."""
    from rayca_amd import lib
    lib.check(product_lib.rayca_hip_selftest())
