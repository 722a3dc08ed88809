"""Phase times of the host-side scene build on the CPU (no GPU needed: the BLAS is built by the host builder).
usage: RAYCA_BUILD_TIMING=1 python tests/host_build_probe.py [atrium|soup] [repeats]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rayca_amd import abi, flatten, scenes
so = os.path.join(ROOT, "tests", "cpp", "_build", "libhost_build_probe.so")
subprocess.run(["g++", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared", "-pthread", "-I" + os.path.join(ROOT, "include"),
                os.path.join(ROOT, "tests", "cpp", "host_build_probe.cpp"), os.path.join(ROOT, "rayca_amd", "csrc", "host_scene.cpp"), "-o", so], check=True)
lib = C.CDLL(so)
which = sys.argv[1] if len(sys.argv) > 1 else "atrium"
desc = flatten(scenes.atrium_scene() if which == "atrium" else scenes.soup_scene(1_000_000))
lib.host_build_probe.argtypes = [C.c_void_p, C.c_uint, C.c_int]
sys.exit(lib.host_build_probe(C.cast(desc.ptr(), C.c_void_p), abi.BUILDER_SAH, int(sys.argv[2]) if len(sys.argv) > 2 else 3))
