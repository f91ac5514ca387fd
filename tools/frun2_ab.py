"""Band-fused nine-point sweep (CEDAR_AMD_FRUN2 = F rows per workgroup, 0 = one launch per row class): relax sweep and
V-cycle time of the resident 2D solver for several grid sizes and run lengths, interleaved on one allocation.
    python tools/frun2_ab.py [n ...]      # default 4096 2048 1024"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import ctypes as C
import numpy as np
from cedar_amd import capi

capi.lib.cedar_amd_solver_time_relax.restype = C.c_float
capi.lib.cedar_amd_solver_time_vcycles.restype = C.c_float
for n in [int(a) for a in sys.argv[1:]] or [4096, 2048, 1024]:
    so, b = capi.gallery("fe2", (n, n))
    x = capi.DeviceArray(b.shape); x.zero()
    s = capi.Solver(so, share_operator=True)
    for rep in range(2):
        for frun in (0, 4, 8, 16, 32):
            os.environ["CEDAR_AMD_FRUN2"] = str(frun)
            os.environ["CEDAR_AMD_NO_GRAPH"] = "1"  # the graph would replay the first setting
            capi.lib.cedar_amd_solver_time_relax(s.h, capi._vp(x), capi._vp(b), 4)
            sw = capi.lib.cedar_amd_solver_time_relax(s.h, capi._vp(x), capi._vp(b), 40) / 40
            print("n=%5d frun2=%2d  sweep %.4f ms  (%.0f GB/s algorithmic at 72 B/DOF)" % (n, frun, sw, 72.0 * n * n / sw / 1e6), flush=True)
    s.close()
