"""profiling helper: level-0 line-xy sweeps of the resident 2D solver at n^2"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ["CEDAR_AMD_NO_GRAPH"] = "1"
if len(sys.argv) > 3:
    os.environ["CEDAR_AMD_LINE_PREFETCH"] = sys.argv[3]
import problems as pb
from cedar_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
so = capi.DeviceArray.from_numpy(pb.aniso9(n, n))
b = capi.DeviceArray.from_numpy(pb.rhs2(n, n))
x = capi.DeviceArray(b.shape)
s = capi.Solver(so, relax="line-xy", share_operator=True)
print("ms per xy sweep", s.time_relax(x, b, reps) / reps)
