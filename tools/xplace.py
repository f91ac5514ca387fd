"""Where do the two modes of the 512^3 relax sweep (5.3 / 6.1 ms) come from?  One solver (operator + solve copy fixed), the
right-hand side fixed, x placed (a) at several offsets inside one large buffer, (b) in several fresh buffers held at once;
then the same for b with x fixed.  Prints the sweep time for each placement and the device addresses.
    python tools/xplace.py [n]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from cedar_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
capi.lib.cedar_amd_solver_time_relax.restype = C.c_float
so, b = capi.gallery("fe3", (n, n, n))
s = capi.Solver(so, share_operator=True)
npts = (n + 2) ** 3


def sweep(xp, bp):
    capi.lib.cedar_amd_solver_time_relax(s.h, C.c_void_p(xp), C.c_void_p(bp), 2)
    return capi.lib.cedar_amd_solver_time_relax(s.h, C.c_void_p(xp), C.c_void_p(bp), 10) / 10


big = capi.DeviceArray((npts + (1 << 26),))
big.zero()
print("operator %#x  b %#x  big %#x" % (so.ptr, b.ptr, big.ptr))
for off in (0, 32, 512, 4096, 1 << 16, 1 << 20, 1 << 24, 1 << 28, (1 << 28) + 4096):
    print("x at big + %10d B: %.3f ms per sweep" % (off, sweep(big.ptr + off, b.ptr)), flush=True)
held = []
for i in range(8):
    x = capi.DeviceArray((npts,))
    x.zero()
    held.append(x)
    print("x in fresh buffer %d at %#x: %.3f ms per sweep" % (i, x.ptr, sweep(x.ptr, b.ptr)), flush=True)
x0 = held[0]
for i in range(1, 8):
    print("b in buffer %d at %#x, x at %#x: %.3f ms per sweep" % (i, held[i].ptr, x0.ptr, sweep(x0.ptr, held[i].ptr)), flush=True)
