"""level-0 relax sweeps of the resident 27-point solver at n^3 (environment passed through: CEDAR_AMD_PSUM, _FRUN, _ILV).
usage: psum_relax.py n reps"""
import os, sys, json
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
os.environ["CEDAR_AMD_NO_GRAPH"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cedar_amd import capi
so, b = capi.gallery("fe3", (n, n, n))
x = capi.DeviceArray(b.shape)
s = capi.Solver(so, share_operator=True)
s.time_relax(x, b, 2)
ms = s.time_relax(x, b, reps) / reps
print(json.dumps({"n": n, "psum": os.environ.get("CEDAR_AMD_PSUM", ""), "frun": os.environ.get("CEDAR_AMD_FRUN", ""),
                  "ms_per_sweep": ms}))
