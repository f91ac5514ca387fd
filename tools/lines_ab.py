"""8192^2 line-xy workload: V-cycle and level-0 sweep time with / without the scan-ordered factor copies
(CEDAR_AMD_LINE_PERM, read when the solver is created), interleaved rounds on two solvers sharing the operator."""
import os, sys, json, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import problems as pb
from cedar_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
os.environ["CEDAR_AMD_NO_GRAPH"] = "1"
so = capi.DeviceArray.from_numpy(pb.aniso9(n, n))
b = capi.DeviceArray.from_numpy(pb.rhs2(n, n))
x = capi.DeviceArray(b.shape)
S = {}
for f in ("1", "0"):
    os.environ["CEDAR_AMD_LINE_PERM"] = f
    S[f] = capi.Solver(so, relax="line-xy", share_operator=True)
res = {"1": [], "0": []}
for rnd in range(5):
    for f in ("1", "0"):
        s = S[f]
        s.time_relax(x, b, 2)
        tr = s.time_relax(x, b, 8) / 8
        tv = s.time_vcycles(x, b, 4) / 4
        res[f].append((tr, tv))
for f in ("1", "0"):
    print(json.dumps({"n": n, "scan_ordered_factors": f, "relax_xy_ms_per_sweep": statistics.median(r[0] for r in res[f]),
                      "vcycle_ms": statistics.median(r[1] for r in res[f])}))
