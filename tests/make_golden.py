"""Generates tests/golden/*.npz with the CPU oracle (run in the build container; the fixtures are
committed because neither the reference nor this script's inputs need to exist on the GPU box).

The reference itself ships no golden images (its integration tests only dump PNGs), and it cannot be
built here, so these vectors pin the ORACLE's behaviour at the time of generation: a regression guard
and a second anchor for the GPU parity tests, not an independent confirmation of the oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402
from rayca_amd import Config, IntegratorStrategy, flatten, scenes  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def rays_for(seed, n, lo, hi):
    """n rays from points on a sphere shell toward points inside [lo,hi]^3 (deterministic)."""
    i = np.arange(n * 6, dtype=np.uint32)
    u = scenes.hash_unit(seed, i).reshape(n, 6)
    o = (u[:, :3] * 2 - 1).astype(np.float32)
    o = o / np.maximum(np.linalg.norm(o, axis=1, keepdims=True), 1e-3) * np.float32(hi * 3)
    tgt = (u[:, 3:] * (hi - lo) + lo).astype(np.float32)
    d = tgt - o
    return np.concatenate([o, d], 1).astype(np.float32)


def main():
    os.makedirs(OUT, exist_ok=True)
    # C1: the Box glTF, 256x256 (rayca-soft/tests/gltf.rs:191-204)
    d = flatten(scenes.box_scene())
    o = ol.OracleScene(d, Config())
    _, flat, _ = o.render(Config(integrator=IntegratorStrategy.Flat), 256, 256)
    _, pt1, _ = o.render(Config(max_depth=1), 256, 256)
    rays = rays_for(11, 512, -0.7, 0.7)
    t, prim, uv, _ = o.trace_rays(rays)
    np.savez_compressed(os.path.join(OUT, "box_256.npz"), flat=flat, pt1=pt1, rays=rays, t=t, prim=prim, uv=uv)
    # Cornell-style room 128x72 and 1k-triangle soup hit records
    d = flatten(scenes.cornell_scene())
    o = ol.OracleScene(d, Config())
    _, flat, _ = o.render(Config(integrator=IntegratorStrategy.Flat), 128, 72)
    _, pt1, _ = o.render(Config(max_depth=1), 128, 72)
    rays = rays_for(12, 512, -1.0, 2.0)
    t, prim, uv, _ = o.trace_rays(rays)
    np.savez_compressed(os.path.join(OUT, "cornell_128x72.npz"), flat=flat, pt1=pt1, rays=rays, t=t, prim=prim, uv=uv)
    d = flatten(scenes.soup_scene(1000, extent=0.12))
    o = ol.OracleScene(d, Config())
    rays = rays_for(13, 2048, -1.0, 1.0)
    t, prim, uv, _ = o.trace_rays(rays)
    np.savez_compressed(os.path.join(OUT, "soup1k_rays.npz"), rays=rays, t=t, prim=prim, uv=uv, order=o.primitive_order())
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
