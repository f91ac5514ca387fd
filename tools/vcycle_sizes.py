"""V(2,1) throughput of the 27-pt gallery::fe problem as a function of the grid size (one GPU, hipGraph replay)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from cedar_amd import capi
for n in [int(a) for a in sys.argv[1:]] or [32, 64, 96, 128, 192, 256, 320, 384, 512, 640, 768]:
    so, b = capi.gallery("fe3", (n, n, n))
    s = capi.Solver(so, share_operator=True)
    x = capi.DeviceArray(b.shape)
    for _ in range(3):
        s.vcycle(x, b)
    k = 20 if n <= 256 else 5
    ms = s.time_vcycles(x, b, k) / k
    print(json.dumps({"n": n, "levels": s.nlevels(), "ms_per_vcycle": ms, "dof_per_s": n ** 3 / ms * 1e3}), flush=True)
    s.close()
    del so, b, x
