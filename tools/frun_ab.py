"""interleaved A/B of the plane-fused run length on ONE solver (CEDAR_AMD_FRUN is read per call), many short rounds:
the card's own drift (cold / warm) is larger than the differences between settings, so settings are compared
round by round, not run by run"""
import os, sys, json, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cedar_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
settings = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else "8,16,32,0".split(","))]
so, b = capi.gallery("fe3", (n, n, n))
x = capi.DeviceArray(b.shape)
os.environ["CEDAR_AMD_NO_GRAPH"] = "1"
s = capi.Solver(so, share_operator=True)
res = {f: [] for f in settings}
for rnd in range(12):
    for f in settings:
        os.environ["CEDAR_AMD_FRUN"] = str(f)
        s.time_relax(x, b, 2)
        res[f].append(s.time_relax(x, b, 6) / 6)
for f in settings:
    v = res[f]
    print(json.dumps({"n": n, "frun": f, "median_ms_per_sweep": statistics.median(v), "min": min(v), "max": max(v)}), flush=True)
base = res[settings[0]]
for f in settings[1:]:
    print("frun %d vs %d: median of the per-round ratios %.4f" % (f, settings[0], statistics.median([a / c for a, c in zip(res[f], base)])))
