"""rayca_amd -- MI355X (gfx950) path-tracing core behind rayca-soft's Scene / draw() surface.

Only what the hot path needs: `csrc/` (hand-written HIP kernels + the C ABI, built into
csrc/librayca_hip.so), the ctypes mirror of that ABI, and the host-side mirror of the reference's
scene/config/renderer types.  There is no CPU fallback anywhere in this package.
"""
from .abi import SceneDesc  # noqa: F401
from .model import (Camera, GgxMaterial, Image, Light, Mesh, Model, Node, PbrMaterial,  # noqa: F401
                    PhongMaterial, Primitive, Scene, Sphere, Texture, TriangleMesh, Trs,
                    create_default_model, flatten, quat_axis_angle)
from .renderer import Config, DeviceScene, IntegratorStrategy, SamplerStrategy, SoftRenderer  # noqa: F401
