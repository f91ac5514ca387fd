"""profiling / experiment helper: level-0 relax sweeps of the resident 27-point solver at n^3.
usage: relax_solver.py n reps ilv whatif   (ilv, whatif -> CEDAR_AMD_ILV / CEDAR_AMD_WHATIF, set before the library loads)"""
import os, sys, json
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
os.environ["CEDAR_AMD_ILV"] = sys.argv[3] if len(sys.argv) > 3 else "0"
os.environ["CEDAR_AMD_WHATIF"] = sys.argv[4] if len(sys.argv) > 4 else "0"
os.environ["CEDAR_AMD_NO_GRAPH"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cedar_amd import capi
so, b = capi.gallery("fe3", (n, n, n))
x = capi.DeviceArray(b.shape)
s = capi.Solver(so, share_operator=True)
ms = s.time_relax(x, b, reps) / reps
print(json.dumps({"n": n, "ilv": os.environ["CEDAR_AMD_ILV"], "whatif": os.environ["CEDAR_AMD_WHATIF"], "ms_per_sweep": ms}))
