"""Host<->device copy rates on this box: pageable vs pinned, and what pinning costs (ms)."""
import ctypes as C, time
import numpy as np
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipHostFree.argtypes = [C.c_void_p]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
H2D, D2H = 1, 2
hip.hipFree(None)
def t(fn):
    t0 = time.perf_counter(); r = fn(); return (time.perf_counter() - t0) * 1e3, r
for mb in (16, 80):
    n = mb << 20
    d = C.c_void_p(); assert hip.hipMalloc(C.byref(d), n) == 0
    for rnd in range(2):
        a = np.ones(n, np.uint8)   # fresh pageable memory, touched
        th, _ = t(lambda: hip.hipMemcpy(d, a.ctypes.data, n, H2D))
        td, _ = t(lambda: hip.hipMemcpy(a.ctypes.data, d, n, D2H))
        print(f"{mb} MiB pageable: H2D {th:6.2f} ms ({n / th / 1e6:5.1f} GB/s)  D2H {td:6.2f} ms ({n / td / 1e6:5.1f} GB/s)", flush=True)
        tr, rc = t(lambda: hip.hipHostRegister(a.ctypes.data, n, 0))
        th, _ = t(lambda: hip.hipMemcpy(d, a.ctypes.data, n, H2D))
        td, _ = t(lambda: hip.hipMemcpy(a.ctypes.data, d, n, D2H))
        tu, _ = t(lambda: hip.hipHostUnregister(a.ctypes.data))
        print(f"{mb} MiB registered in place (rc {rc}): register {tr:6.2f} ms, H2D {th:6.2f} ms ({n / th / 1e6:5.1f} GB/s), D2H {td:6.2f} ms ({n / td / 1e6:5.1f} GB/s), unregister {tu:6.2f} ms", flush=True)
    p = C.c_void_p()
    ta, rc = t(lambda: hip.hipHostMalloc(C.byref(p), n, 0))
    tt, _ = t(lambda: C.memset(p, 1, n))
    th, _ = t(lambda: hip.hipMemcpy(d, p, n, H2D))
    td, _ = t(lambda: hip.hipMemcpy(p, d, n, D2H))
    tf, _ = t(lambda: hip.hipHostFree(p))
    print(f"{mb} MiB hipHostMalloc (rc {rc}): alloc {ta:6.2f} ms, first touch {tt:6.2f} ms, H2D {th:6.2f} ms ({n / th / 1e6:5.1f} GB/s), D2H {td:6.2f} ms ({n / td / 1e6:5.1f} GB/s), free {tf:6.2f} ms", flush=True)
    hip.hipFree(d)
