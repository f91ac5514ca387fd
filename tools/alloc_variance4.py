"""relax sweep on operators allocated through the HIP virtual-memory API (tools/vmm_probe.cpp) vs hipMalloc"""
import os, sys, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cedar_amd import capi
n = 512
os.environ["CEDAR_AMD_NO_GRAPH"] = "1"
probe = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "libvmmprobe.so"))
probe.vmm_alloc.restype = C.c_void_p
probe.vmm_alloc.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t]
so, b = capi.gallery("fe3", (n, n, n))
x = capi.DeviceArray(b.shape)
npts = (n + 2) ** 3
nbytes = 14 * npts * 8


class View:
    def __init__(self, ptr, shape):
        self.ptr, self.shape = ptr, shape

    def data_ptr(self):
        return self.ptr


def sweep(v):
    s = capi.Solver(v, share_operator=True)
    s.time_relax(x, b, 2)
    ms = s.time_relax(x, b, 6) / 6
    s.close()
    return round(ms, 4)


probe.vmm_free.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
import statistics
res = {}
for rnd in range(int(os.environ.get("ROUNDS", "5"))):
    methods = (("hipMalloc", -1, 0), ("vmm one handle", 0, 0), ("vmm 4 GB chunks", 4 << 30, 0), ("vmm 1 GB chunks", 1 << 30, 0),
               ("vmm 256 MB chunks", 256 << 20, 0), ("vmm 64 MB chunks", 64 << 20, 0), ("vmm 2 MB chunks", 2 << 20, 0))
    if os.environ.get("METHODS") == "short":
        methods = (("hipMalloc", -1, 0), ("vmm 1 GB chunks", 1 << 30, 0), ("vmm plane-sized chunks", (npts * 8 + 4095) // 4096 * 4096, 0),
                   ("vmm 512 MB chunks", 512 << 20, 0))
    for how, chunk, align in methods:
        if chunk < 0:
            so2 = capi.DeviceArray(so.shape)
            so2.copy_from(so)
            ms = sweep(so2)
            so2.free()
        else:
            p = probe.vmm_alloc(nbytes, chunk, align)
            if not p:
                continue
            capi.lib.cedar_amd_memcpy_d2d(p, so.ptr, nbytes)
            ms = sweep(View(p, so.shape))
            probe.vmm_free(p, nbytes, chunk)
        res.setdefault(how, []).append(ms)
        print(json.dumps({"round": rnd, "how": how, "ms_per_sweep": ms}), flush=True)
for how, v in res.items():
    print("%-20s median %.3f  min %.3f  max %.3f   %s" % (how, statistics.median(v), min(v), max(v), v))
