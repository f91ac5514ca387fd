"""27-pt residual at n^3: own-row operator loads cached vs non-temporal (CEDAR_AMD_RESID_NT)"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cedar_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
so, b = capi.gallery("fe3", (n, n, n))
K = capi.Kernels()
g = (n + 2,) * 3
x, r = capi.DeviceArray(g), capi.DeviceArray(g)
for nt in (0, 1, 0, 1):
    os.environ["CEDAR_AMD_RESID_NT"] = str(nt)
    for _ in range(3):
        K.residual3(so, b, x, r)
    capi.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        K.residual3(so, b, x, r)
    capi.sync()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print(json.dumps({"n": n, "resid_nt": nt, "ms": ms, "alg_TBps": 136.0 * n ** 3 / ms / 1e9}), flush=True)
