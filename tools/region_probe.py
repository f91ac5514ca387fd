"""Is device memory uniformly fast?  K buffers of SIZE GB held at once; for each, the time of a device-to-device copy of its
first half onto its second half (HIP events through the library), three times.
    python tools/region_probe.py [K=12] [SIZE_GB=18]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from cedar_amd import capi

K = int(sys.argv[1]) if len(sys.argv) > 1 else 12
GB = float(sys.argv[2]) if len(sys.argv) > 2 else 18.0
n = int(GB * 2 ** 30 / 8)
half = n // 2
bufs = []
for k in range(K):
    d = capi.DeviceArray((n,))
    d.zero()
    bufs.append(d)
capi.sync()
for rep in range(2):
    for k, d in enumerate(bufs):
        ts = []
        for _ in range(3):
            capi.sync()
            t0 = time.perf_counter()
            capi.lib.cedar_amd_memcpy_d2d(C.c_void_p(d.ptr + 8 * half), C.c_void_p(d.ptr), C.c_size_t(8 * half))
            capi.sync()
            ts.append(time.perf_counter() - t0)
        t = min(ts)
        print("buffer %2d at %#x: copy of %.1f GB in %.3f ms = %.2f TB/s read+write" % (k, d.ptr, half * 8 / 2 ** 30, t * 1e3, 2 * half * 8 / t / 1e12), flush=True)
