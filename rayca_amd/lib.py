"""Loader for librayca_hip.so (the product library).  There is no CPU fallback: if the HIP library
has not been built (or cannot be loaded) every entry point raises -- the product path must fail
loudly rather than silently run somewhere else."""
from __future__ import annotations

import ctypes as C
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RAYCA_HIP_LIB") or os.path.join(_HERE, "csrc", "librayca_hip.so")
_lib = None


class RaycaError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"rayca_hip error {code}: {message}")
        self.code = code


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RaycaError(abi.ERR_NO_DEVICE,
                             f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`"
                             " (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # PyTorch-ROCm ships its own libamdhip64.so.7 / libhsa-runtime64; the system ROCm this library links has the
        # same SONAMEs.  Whichever is loaded first serves both, and torch only finds the GPU with its own pair, so
        # when torch is installed it goes first (bench.py and the multi-GPU path use torch for streams and RCCL).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = abi.bind_product_signatures(C.CDLL(LIB_PATH))
    return _lib


def last_error() -> str:
    buf = C.create_string_buffer(1024)
    load().rayca_hip_last_error(buf, len(buf))
    return buf.value.decode("utf-8", "replace")


def check(rc: int):
    if rc != abi.OK:
        raise RaycaError(rc, last_error())
