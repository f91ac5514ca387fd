"""level-0 residual / restrict / interp_add launch times of the resident 27-point solver at n^3 (environment passed through)"""
import os, sys, json
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cedar_amd import capi
so, b = capi.gallery("fe3", (n, n, n))
x = capi.DeviceArray(b.shape)
s = capi.Solver(so, share_operator=True)
out = {"n": n, "env": {k: v for k, v in os.environ.items() if k.startswith("CEDAR_AMD_")}}
for op in sys.argv[2:] or ["residual"]:
    s.time_op(x, b, op, 3)
    out[op] = s.time_op(x, b, op, 10) / 10
print(json.dumps(out))
