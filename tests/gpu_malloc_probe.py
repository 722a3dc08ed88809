"""What hipMalloc / hipFree cost on this box: fresh memory vs. a size the process has already held (ms per call)."""
import ctypes as C, time
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
hip.hipFree(None)
def t(fn):
    t0 = time.perf_counter(); r = fn(); return (time.perf_counter() - t0) * 1e3, r
def malloc(n):
    p = C.c_void_p(); rc = hip.hipMalloc(C.byref(p), n); assert rc == 0, rc; return p
for rnd in range(2):
    for mb in (1, 8, 32, 128, 512):
        n = mb << 20
        ta, p = t(lambda: malloc(n))
        tm, _ = t(lambda: (hip.hipMemset(p, 0, n), hip.hipDeviceSynchronize()))
        tf, _ = t(lambda: hip.hipFree(p))
        print(f"round {rnd}: {mb:4d} MiB  hipMalloc {ta:7.3f} ms   first touch (memset+sync) {tm:7.3f} ms   hipFree {tf:7.3f} ms", flush=True)
# twelve small allocations against one of the same total
ta, ps = t(lambda: [malloc(8 << 20) for _ in range(12)])
tf, _ = t(lambda: [hip.hipFree(p) for p in ps])
print(f"12 x 8 MiB: hipMalloc {ta:.3f} ms, hipFree {tf:.3f} ms")
ta, p = t(lambda: malloc(96 << 20)); tf, _ = t(lambda: hip.hipFree(p))
print(f"1 x 96 MiB: hipMalloc {ta:.3f} ms, hipFree {tf:.3f} ms")
