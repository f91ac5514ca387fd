"""which levels should carry the row-interleaved solve copy?  V-cycle and level-0 relax time for CEDAR_AMD_ILV thresholds,
several fresh allocations each (placement spread)."""
import os, sys, json, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cedar_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["0", "1", "300", "100"]
trials = int(sys.argv[3]) if len(sys.argv) > 3 else 3
res = {m: [] for m in modes}
for t in range(trials):
    for m in modes:
        os.environ["CEDAR_AMD_ILV"] = m
        so, b = capi.gallery("fe3", (n, n, n))
        x = capi.DeviceArray(b.shape)
        s = capi.Solver(so, share_operator=True)
        s.time_relax(x, b, 2)
        tr = min(s.time_relax(x, b, 6) / 6 for _ in range(2))
        s.time_vcycles(x, b, 2)
        tv = min(s.time_vcycles(x, b, 5) / 5 for _ in range(2))
        res[m].append((tr, tv))
        print(json.dumps({"n": n, "ilv": m, "trial": t, "relax_ms": tr, "vcycle_ms": tv}), flush=True)
        s.close(); so.free(); b.free(); x.free()
for m in modes:
    print("ilv %s: relax median %.3f, vcycle median %.3f min %.3f" % (m, statistics.median(r[0] for r in res[m]),
          statistics.median(r[1] for r in res[m]), min(r[1] for r in res[m])))
