import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from cedar_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
op = os.environ.get("CEDAR_AMD_TOOL_OP", "fe3")  # fe3 (27-point) or poisson3 (7-point fine level, basic-3d-ser/poisson.cc)
so, b = capi.gallery(op, (n, n, n))
s = capi.Solver(so, share_operator=True, num_levels=2 if len(sys.argv) > 2 else -1)
x = capi.DeviceArray(b.shape)
for _ in range(2): s.vcycle(x, b)
capi.sync()
ms = s.time_vcycles(x, b, 5) / 5
print("vcycle %.3f ms" % ms, "x_l2 %.15e" % capi.l2norm(x))
