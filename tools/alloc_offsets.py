"""relax sweep time against the placement of x and b relative to the operator: x and b are views into one big
buffer at controlled byte offsets; the operator and the solver's own arrays stay where they are"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cedar_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
os.environ["CEDAR_AMD_NO_GRAPH"] = "1"
so, b0 = capi.gallery("fe3", (n, n, n))
s = capi.Solver(so, share_operator=True)
npts = (n + 2) ** 3
big = capi.DeviceArray((3 * npts + (1 << 22),))   # room for x and b plus 32 MB of play


class View:
    def __init__(self, ptr):
        self.ptr = ptr


base = (big.ptr + 4095) // 4096 * 4096
print(json.dumps({"so": hex(so.ptr), "so_mod_2M": so.ptr % (2 << 20), "base": hex(base), "plane_bytes": npts * 8}), flush=True)
for offx, offb in ((0, 0), (0, 256), (0, 1024), (0, 4096), (0, 16384), (0, 65536), (0, 1 << 20), (256, 0), (4096, 0), (65536, 0),
                   (1 << 20, 0), (0, 0), (4096, 4096 * 3), (65536, 65536 * 3), (1 << 21, 1 << 22)):
    x = View(base + offx)
    b = View(base + npts * 8 + (1 << 23) + offb)
    s.time_relax(x, b, 2)
    ms = [s.time_relax(x, b, 6) / 6 for _ in range(2)]
    print(json.dumps({"offx": offx, "offb": offb, "x_mod_2M": x.ptr % (2 << 20), "ms_per_sweep": [round(v, 4) for v in ms]}), flush=True)
