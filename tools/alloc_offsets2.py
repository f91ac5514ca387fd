"""relax sweep time against the byte offset of the OPERATOR inside one big allocation (same physical neighbourhood, shifted start)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cedar_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
os.environ["CEDAR_AMD_NO_GRAPH"] = "1"
so, b = capi.gallery("fe3", (n, n, n))
x = capi.DeviceArray(b.shape)
npts = (n + 2) ** 3
big = capi.DeviceArray((14 * npts + (1 << 24),))   # operator + 128 MB of play


class View:
    def __init__(self, ptr, shape):
        self.ptr, self.shape = ptr, shape

    def data_ptr(self):
        return self.ptr


base = (big.ptr + (2 << 20) - 1) // (2 << 20) * (2 << 20)
print(json.dumps({"base": hex(base), "plane_bytes": npts * 8, "plane_mod_2M": (npts * 8) % (2 << 20)}), flush=True)
offs = [0, 256, 1024, 4096, 16384, 65536, 262144, 1 << 20, 2 << 20, 3 << 20, 4 << 20, 6 << 20, 8 << 20, 12 << 20, 16 << 20, 24 << 20, 32 << 20, 48 << 20, 64 << 20, 0]
for off in offs:
    v = View(base + off, so.shape)
    capi.lib.cedar_amd_memcpy_d2d(v.ptr, so.ptr, 14 * npts * 8)
    s = capi.Solver(v, share_operator=True)
    s.time_relax(x, b, 2)
    ms = s.time_relax(x, b, 6) / 6
    print(json.dumps({"off": off, "off_MB": off / (1 << 20), "ms_per_sweep": round(ms, 4)}), flush=True)
    s.close()
