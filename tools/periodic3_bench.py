"""3D periodic boundary conditions: V-cycle time of the device-resident solver against the Dirichlet solver on the same
27-point operator size (the periodic sweep runs one launch per colour plus ghost refreshes instead of the fused passes).
    python tools/periodic3_bench.py [n ...]     # default 128 256"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import problems as pb
from cedar_amd import capi

capi.lib.cedar_amd_solver_time_vcycles.restype = C.c_float
for n in [int(a) for a in sys.argv[1:]] or [128, 256]:
    for per in ((0, 0, 0), (0, 0, 1), (1, 0, 0), (1, 1, 1)):
        so = pb.periodic_random_op3(n, n, n, 14, per, 5)
        b = pb.periodic_rhs3(n, n, n, per)
        s = capi.Solver(so, ibc=pb.ibc3_of(per))
        xd, bd = capi.DeviceArray.from_numpy(np.zeros_like(b)), capi.DeviceArray.from_numpy(b)
        s.vcycle(xd, bd)
        capi.sync()
        ms = capi.lib.cedar_amd_solver_time_vcycles(s.h, capi._vp(xd), capi._vp(bd), 5) / 5
        h = s.solve(b, np.zeros_like(b))
        print("n=%4d periodic %s: V-cycle %7.3f ms, %d levels, %d cycles to %.1e" % (n, per, ms, s.nlevels(), len(h) - 1, h[-1]), flush=True)
        s.close()
