import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from cedar_amd import capi
n = 512
so, b = capi.gallery("fe3", (n, n, n))
capi.sync(); t0 = time.time()
s = capi.Solver(so, share_operator=True)
capi.sync(); t1 = time.time()
A1 = s.array(3, "A")
print("setup %.1f ms" % ((t1 - t0) * 1e3), "level-3 operator checksum %.17g" % float(np.sum(A1 * np.arange(A1.size).reshape(A1.shape) % 7)))
