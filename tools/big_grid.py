"""largest single-GPU case: 27-pt gallery::fe at n^3 (default 1024^3: ~215 GB of the 288 GB HBM), device-resident
hierarchy, a few V-cycles; prints set-up time, cycle time, DOF/s and the residual reduction"""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from cedar_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
so, b = capi.gallery("fe3", (n, n, n))
capi.sync()
t0 = time.perf_counter()
s = capi.Solver(so, share_operator=True, max_iter=4, tol=1e-30)
capi.sync()
t_setup = time.perf_counter() - t0
x = capi.DeviceArray(b.shape)
s.vcycle(x, b)
ms = s.time_vcycles(x, b, 3) / 3
nsw = 6
s.time_relax(x, b, 2)
rl = s.time_relax(x, b, nsw) / nsw
x2 = capi.DeviceArray(b.shape)
h = s.solve(b, x2)
print(json.dumps({"n": n, "levels": s.nlevels(), "setup_s": t_setup, "ms_per_vcycle": ms, "dof_per_s": n ** 3 / ms * 1e3,
                  "relax_sweep_ms": rl, "relax_alg_TBps": 136.0 * n ** 3 / rl / 1e9, "history": [float(v) for v in h]}), flush=True)
