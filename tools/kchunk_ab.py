"""Plane-chunked order of the plane-fused 27-point sweep (CEDAR_AMD_KCHUNK = second-parity planes per chunk, 0 = all
first-parity planes, then all second-parity planes) x run length (CEDAR_AMD_FRUN), interleaved on one allocation; the
sweeps are compared bit for bit.    python tools/kchunk_ab.py [n]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from cedar_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
capi.lib.cedar_amd_solver_time_relax.restype = C.c_float
so, b = capi.gallery("fe3", (n, n, n))
s = capi.Solver(so, share_operator=True)
x = capi.DeviceArray(b.shape)
ref = None
for rep in range(2):
    for frun in (8, 4, 2):
        for kc in (0, 16, 8, 4, 2):
            os.environ["CEDAR_AMD_FRUN"] = str(frun)
            os.environ["CEDAR_AMD_KCHUNK"] = str(kc)
            x.zero()
            capi.lib.cedar_amd_solver_time_relax(s.h, capi._vp(x), capi._vp(b), 2)
            h = x.numpy()
            if ref is None:
                ref = h
            same = np.array_equal(h, ref)
            ms = capi.lib.cedar_amd_solver_time_relax(s.h, capi._vp(x), capi._vp(b), 10) / 10
            print("frun=%d kchunk=%2d  sweep %.3f ms  %s" % (frun, kc, ms, "same bits" if same else "DIFFERENT"), flush=True)
