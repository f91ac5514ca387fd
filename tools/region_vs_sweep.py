"""Does the copy bandwidth of the buffer that holds the operator predict the relax sweep time?  Eight 27-point operators
(15 GB each, Cedar layout, no solve copy: CEDAR_AMD_ILV=0, operator shared with the solver) held at once; for each the
half-onto-half copy time of its buffer and the sweep time of a solver on it (one x, one b for all).
    python tools/region_vs_sweep.py"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ["CEDAR_AMD_ILV"] = "0"
from cedar_amd import capi

capi.lib.cedar_amd_solver_time_relax.restype = C.c_float
n = 512
ops = []
b0 = None
for k in range(8):
    so, b = capi.gallery("fe3", (n, n, n), with_rhs=(k == 0))
    if k == 0:
        b0 = b
    ops.append(so)
x = capi.DeviceArray(b0.shape)
x.zero()
for rep in range(2):
    for k, so in enumerate(ops):
        half = so.size // 2
        tmp = capi.DeviceArray((half,))  # copy target elsewhere: the operator must stay intact
        ts = []
        for _ in range(3):
            capi.sync(); t0 = time.perf_counter()
            capi.lib.cedar_amd_memcpy_d2d(C.c_void_p(tmp.ptr), C.c_void_p(so.ptr), C.c_size_t(8 * half))
            capi.sync(); ts.append(time.perf_counter() - t0)
        tmp.free()
        s = capi.Solver(so, share_operator=True)
        capi.lib.cedar_amd_solver_time_relax(s.h, capi._vp(x), capi._vp(b0), 2)
        ms = capi.lib.cedar_amd_solver_time_relax(s.h, capi._vp(x), capi._vp(b0), 8) / 8
        s.close()
        print("operator %d at %#x: read of its first half %.3f ms (%.2f TB/s)   relax sweep %.3f ms" % (k, so.ptr, min(ts) * 1e3, half * 8 / min(ts) / 1e12, ms), flush=True)
