"""where does the x-line kernel spend its time?  level-0 line-x sweeps at n^2 with parts of the kernel switched off
(CEDAR_AMD_LINE_DBG: 0 whole kernel, 1 no tridiagonal solve, 2 no right-hand side, 3 neither)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ["CEDAR_AMD_NO_GRAPH"] = "1"
import problems as pb
from cedar_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
so = capi.DeviceArray.from_numpy(pb.aniso9(n, n))
b = capi.DeviceArray.from_numpy(pb.rhs2(n, n))
x = capi.DeviceArray(b.shape)
for pf in ("0", "1"):
    os.environ["CEDAR_AMD_LINE_PERM"] = pf
    s = capi.Solver(so, relax="line-x", share_operator=True)
    for d in ("0", "1", "2", "3"):
        os.environ["CEDAR_AMD_LINE_DBG"] = d
        s.time_relax(x, b, 2)
        print(json.dumps({"n": n, "scan_ordered_factors": pf, "dbg": d, "ms_per_x_sweep": s.time_relax(x, b, 8) / 8}), flush=True)
    s.close()
