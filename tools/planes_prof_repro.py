"""the multi-stream plane-relaxation path (CEDAR_AMD_PLANE_BATCH=0) for a run under rocprofv3: writes /proc/self/maps
next to the log first, so that the raw addresses of a crash report can be mapped to libraries afterwards.
usage: rocprofv3 --kernel-trace --stats ... -- python3 tools/planes_prof_repro.py MAPS_OUT [n] [streams]"""
import os, sys
os.environ["CEDAR_AMD_PLANE_BATCH"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import ctypes
import numpy as np
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
os.environ["CEDAR_AMD_PLANE_STREAMS"] = sys.argv[3] if len(sys.argv) > 3 else "8"
from cedar_amd import capi
import problems as pb
so = pb.diag_diffusion3(n, n, n, 1.0, 1e-2, 1e-4)
b = pb.rhs3(n, n, n)
s = capi.Solver(so, relax="plane-xy")
xd, bd = capi.DeviceArray.from_numpy(np.zeros_like(b)), capi.DeviceArray.from_numpy(b)
s.vcycle(xd, bd)
capi.sync()
open(sys.argv[1], "w").write(open("/proc/self/maps").read())
capi.lib.cedar_amd_solver_time_vcycles.restype = ctypes.c_float
ms = capi.lib.cedar_amd_solver_time_vcycles(s.h, capi._vp(xd), capi._vp(bd), 6) / 6
print("plane-xy multi-stream n=%d: %.2f ms per V-cycle" % (n, ms), flush=True)
