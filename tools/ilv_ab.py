"""Row-interleaved solve copy of the 27-point operator (CEDAR_AMD_ILV) against the Cedar layout: bit-identity of a
solve at a small size, then relax-sweep / V-cycle times at n^3 over several fresh allocations of either layout."""
import os, sys, json, statistics
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cedar_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 3

def solve_small(mode):
    os.environ["CEDAR_AMD_ILV"] = str(mode)
    so, b = capi.gallery("fe3", (44, 37, 30))
    s = capi.Solver(so, max_iter=6)
    x = capi.DeviceArray(b.shape)
    h = s.solve(b, x)
    xs = x.numpy().copy()
    s.close()
    return h, xs

h0, x0 = solve_small(0)
h1, x1 = solve_small(1)
print("small solve identical:", bool(np.array_equal(h0, h1) and np.array_equal(x0, x1)), h0[-1], h1[-1], flush=True)

res = {0: [], 1: []}
for t in range(trials):
    for mode in (0, 1):
        os.environ["CEDAR_AMD_ILV"] = str(mode)
        so, b = capi.gallery("fe3", (n, n, n))
        x = capi.DeviceArray(b.shape)
        s = capi.Solver(so, share_operator=True)
        s.time_relax(x, b, 2)
        tr = min(s.time_relax(x, b, 6) / 6 for _ in range(3))
        s.time_vcycles(x, b, 2)
        tv = min(s.time_vcycles(x, b, 5) / 5 for _ in range(2))
        res[mode].append((tr, tv))
        print(json.dumps({"n": n, "ilv": mode, "trial": t, "relax_ms_per_sweep": tr, "vcycle_ms": tv}), flush=True)
        s.close(); so.free(); b.free(); x.free()
for mode in (0, 1):
    print("ilv %d: relax median %.3f ms (min %.3f max %.3f), vcycle median %.3f ms" % (
        mode, statistics.median(r[0] for r in res[mode]), min(r[0] for r in res[mode]), max(r[0] for r in res[mode]),
        statistics.median(r[1] for r in res[mode])))
