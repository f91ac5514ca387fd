import os, sys
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
from cedar_amd import capi
n = 512
K = capi.Kernels()
so, _ = capi.gallery("fe3", (n, n, n), with_rhs=False)
nc = (n - 1) // 2 + 1
gc = (nc + 2,) * 3
ci = capi.DeviceArray((26,) + gc); ci.zero()
K.setup_interp3(so, ci)
soc = capi.DeviceArray((14,) + gc); soc.zero()
for _ in range(2):
    K.galerkin3(so, soc, ci)
capi.sync()
