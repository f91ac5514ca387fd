"""relax-sweep time at 512^3 as a function of the plane-fused run length (CEDAR_AMD_FRUN; 0 = four launches)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cedar_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
so, b = capi.gallery("fe3", (n, n, n))
x = capi.DeviceArray(b.shape)
for frun in ([0, 256, 128, 64, 32, 16, 8, 0] if n >= 512 else [0, 64, 32, 16, 8, 4, 0]):
    os.environ["CEDAR_AMD_FRUN"] = str(frun)
    os.environ["CEDAR_AMD_NO_GRAPH"] = "1"
    s = capi.Solver(so, share_operator=True)
    s.time_relax(x, b, 4)
    ms = s.time_relax(x, b, 20) / 20
    print(json.dumps({"n": n, "frun": frun, "ms_per_sweep": ms, "alg_TBps": 136.0 * n ** 3 / ms / 1e9}), flush=True)
    s.close()
