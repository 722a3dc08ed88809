"""rayca_hip_render_multi: several devices behind the C ABI.  The one-GPU box can rehearse everything but
the RCCL exchange itself: several scene handles on ONE device with the peer-copy transport exercise the row split, the
per-device threads, the gather layout and the de-interleave kernel; RCCL's own path needs distinct devices."""
import numpy as np
import pytest

from rayca_amd import Config, DeviceScene, IntegratorStrategy, abi, flatten, lib, scenes
from rayca_amd.lib import RaycaError
from rayca_amd.renderer import render_multi


def test_rccl_can_be_opened(product_lib):
    """no GPU needed: the library opens librccl at first use and reports what it found"""
    rc = product_lib.rayca_hip_rccl_status()
    assert rc == abi.OK, lib.last_error()


@pytest.mark.gpu
@pytest.mark.parametrize("parts,size,band", [(1, (320, 180), 8), (2, (320, 181), 8), (3, (257, 99), 4), (8, (640, 360), 8), (5, (64, 7), 8)])
def test_multi_equals_single(gpu, parts, size, band):
    w, h = size
    desc = flatten(scenes.cornell_scene())
    handles = [DeviceScene(desc, Config(), builder=abi.BUILDER_SAH) for _ in range(parts)]
    for cfg in (Config(integrator=IntegratorStrategy.Flat), Config(max_depth=2, seed=3)):
        want, _, st1 = handles[0].render(cfg, w, h, want_f32=False, collect_stats=True)
        got, st = render_multi(handles, cfg, w, h, band_rows=band, gather=abi.GATHER_PEER_COPY, collect_stats=True)
        assert np.array_equal(got, want)
        assert sum(s["rows_rendered"] for s in st) == h                                 # (a part with no rows reports 0)
        for k in ("rays_primary", "rays_shadow", "rays_bounce", "hits_shaded"):
            assert sum(s[k] for s in st) == st1[k], k
        again, _ = render_multi(handles, cfg, w, h, band_rows=band, gather=abi.GATHER_PEER_COPY)     # state is reused
        assert np.array_equal(again, want)
    for hd in handles:
        hd.close()


@pytest.mark.gpu
def test_multi_argument_errors(gpu):
    desc = flatten(scenes.box_scene())
    a, b = DeviceScene(desc, Config()), DeviceScene(desc, Config())
    cfg = Config(integrator=IntegratorStrategy.Flat)
    with pytest.raises(RaycaError) as e:
        render_multi([a, b], cfg, 64, 64, gather=abi.GATHER_RCCL)       # two parts on one device: RCCL refuses that, so do we
    assert e.value.code == abi.ERR_BAD_ARG
    with pytest.raises(RaycaError) as e:
        render_multi([a, a], cfg, 64, 64, gather=abi.GATHER_PEER_COPY)
    assert e.value.code == abi.ERR_BAD_ARG
    with pytest.raises(RaycaError) as e:
        render_multi([a, b], cfg, 64, 64, gather=7)
    assert e.value.code == abi.ERR_BAD_ARG
    one, _ = render_multi([a], cfg, 64, 64, gather=abi.GATHER_RCCL)       # a single part needs no exchange with either transport
    assert np.array_equal(one, a.render(cfg, 64, 64, want_f32=False)[0])
