"""Which part of the stack leaves garbage behind hipMemsetAsync inside the V-cycle (profiles/r01_memset_under_torch_runtime.log)?
argv: [torch] [nograph].  CEDAR_AMD_DEBUG_MEMSET=1 makes the cycle clear coarse x with hipMemsetAsync again.  Prints which
libamdhip64 the process mapped, the runtime version it reports, and the non-zero ghost cells after a 10-cycle solve."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
os.environ["CEDAR_AMD_DEBUG_MEMSET"] = "1"
if "nograph" in sys.argv:
    os.environ["CEDAR_AMD_NO_GRAPH"] = "1"
if "torch" in sys.argv:
    import torch  # noqa: F401
import problems as pb
from cedar_amd import capi
maps = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l})
hip = C.CDLL(maps[0])
v = C.c_int(0)
hip.hipRuntimeGetVersion(C.byref(v))
print("argv", sys.argv[1:], "| libamdhip64 mapped:", maps, "| hipRuntimeGetVersion", v.value, flush=True)
nx, ny = 130, 77
so = pb.aniso9(nx, ny); b = pb.rhs2(nx, ny)
tot = 0
for relax in ("point", "line-xy"):
    s = capi.Solver(so, relax=relax, nrelax_pre=2, nrelax_post=1)
    x = np.zeros_like(b)
    s.solve(b, x)
    n0 = int(np.count_nonzero(x[:, -1]))
    msg = ["%s: level 0 x E ghost nonzero %d" % (relax, n0)]
    tot += n0
    for l in range(1, s.nlevels()):
        a = s.array(l, "x")[0]
        gh = [int(np.count_nonzero(a[:, 0])), int(np.count_nonzero(a[:, -1])), int(np.count_nonzero(a[0, :])), int(np.count_nonzero(a[-1, :]))]
        tot += sum(gh)
        if any(gh):
            msg.append("L%d ghosts WESN %s sample %s" % (l, gh, [hex(int(np.float64(t).view(np.uint64))) for t in a[1:3, -1]]))
    print("  " + " | ".join(msg), flush=True)
    s.close()
print("TOTAL non-zero ghost cells:", tot)
