import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from cedar_amd import capi
n = 512
so, b = capi.gallery("fe3", (n, n, n))
K = capi.Kernels(); g = (n + 2,) * 3
sor, x, r = capi.DeviceArray((2,) + g), capi.DeviceArray(g), capi.DeviceArray(g)
K.setup_recip3(so, sor)
for i in range(2): K.relax3(so, b, x, sor, i & 1); K.residual3(so, b, x, r)
capi.sync(); t0 = time.time()
for i in range(6): K.relax3(so, b, x, sor, i & 1)
capi.sync(); t1 = time.time()
for i in range(6): K.residual3(so, b, x, r)
capi.sync(); t2 = time.time()
print("relax %.3f ms  residual %.3f ms" % ((t1 - t0) / 6 * 1e3, (t2 - t1) / 6 * 1e3))
