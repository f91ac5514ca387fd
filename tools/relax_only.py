"""profiling helper: N relax sweeps (and residuals) of the 27-pt operator at n^3, nothing else timed"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from cedar_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
so, b = capi.gallery("fe3", (n, n, n))
K = capi.Kernels()
g = (n + 2,) * 3
sor, x, r = capi.DeviceArray((2,) + g), capi.DeviceArray(g), capi.DeviceArray(g)
K.setup_recip3(so, sor)
for i in range(reps):
    K.relax3(so, b, x, sor, i & 1)
    K.residual3(so, b, x, r)
capi.sync()
print("done", n, reps)
