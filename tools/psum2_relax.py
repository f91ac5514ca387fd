"""level-0 relax sweeps of the resident 2D nine-point solver at n^2 (environment passed through: CEDAR_AMD_PSUM, _FRUN2)"""
import os, sys, json
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import problems as pb
from cedar_amd import capi
so = capi.DeviceArray.from_numpy(pb.varcoef9(n, n))
b = capi.DeviceArray.from_numpy(pb.rhs2(n, n))
x = capi.DeviceArray(b.shape)
s = capi.Solver(so, share_operator=True)
s.time_relax(x, b, 4)
ms = s.time_relax(x, b, 40) / 40
for _ in range(3):
    s.vcycle(x, b)
capi.sync()
vc = s.time_vcycles(x, b, 10) / 10
print(json.dumps({"n": n, "psum": os.environ.get("CEDAR_AMD_PSUM", ""), "frun2": os.environ.get("CEDAR_AMD_FRUN2", ""),
                  "ms_per_sweep": ms, "ms_per_vcycle": vc}))
