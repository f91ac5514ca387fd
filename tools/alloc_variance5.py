"""is the slow, reproducible sweep on ONE contiguous physical handle a matter of the plane stride?  ns per unknown of the relax sweep
for neighbouring grid sizes, operator in a single-handle VMM allocation (deterministic relative placement of the 14 planes)"""
import os, sys, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cedar_amd import capi
os.environ["CEDAR_AMD_NO_GRAPH"] = "1"
probe = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "libvmmprobe.so"))
probe.vmm_alloc.restype = C.c_void_p
probe.vmm_alloc.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t]
probe.vmm_free.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]


class View:
    def __init__(self, ptr, shape):
        self.ptr, self.shape = ptr, shape

    def data_ptr(self):
        return self.ptr


for n in (496, 504, 508, 510, 512, 514, 516, 520, 528, 544):
    so, b = capi.gallery("fe3", (n, n, n))
    x = capi.DeviceArray(b.shape)
    npts = (n + 2) ** 3
    nbytes = 14 * npts * 8
    out = []
    for rep in range(2):
        p = probe.vmm_alloc(nbytes, 0, 0)
        capi.lib.cedar_amd_memcpy_d2d(p, so.ptr, nbytes)
        s = capi.Solver(View(p, so.shape), share_operator=True)
        s.time_relax(x, b, 2)
        ms = s.time_relax(x, b, 6) / 6
        s.close()
        probe.vmm_free(p, nbytes, 0)
        out.append(round(ms * 1e6 / n ** 3, 5))
    s = capi.Solver(so, share_operator=True)
    s.time_relax(x, b, 2)
    hm = s.time_relax(x, b, 6) / 6 * 1e6 / n ** 3
    s.close()
    print(json.dumps({"n": n, "plane_bytes_mod_64K": (npts * 8) % 65536, "ns_per_dof_one_handle": out, "ns_per_dof_hipMalloc": round(hm, 5)}), flush=True)
    so.free(); b.free(); x.free()
