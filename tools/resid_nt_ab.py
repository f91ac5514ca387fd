"""interleaved A/B of the non-temporal operator loads of the 27-point residual (CEDAR_AMD_RESID_NT, read per call) on ONE
solver: V-cycles without graph replay, many short rounds (the residual is 4 of the 26 ms of a cycle)
    python tools/resid_nt_ab.py [n]"""
import os, sys, json, statistics
os.environ["CEDAR_AMD_NO_GRAPH"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cedar_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
so, b = capi.gallery("fe3", (n, n, n))
x = capi.DeviceArray(b.shape)
s = capi.Solver(so, share_operator=True)
res = {"0": [], "1": []}
for rnd in range(10):
    for f in ("0", "1"):
        os.environ["CEDAR_AMD_RESID_NT"] = f
        s.time_vcycles(x, b, 1)
        res[f].append(s.time_vcycles(x, b, 4) / 4)
for f in ("0", "1"):
    v = res[f]
    print(json.dumps({"n": n, "resid_nt": int(f), "median_ms_per_vcycle": statistics.median(v), "min": min(v), "max": max(v)}), flush=True)
print("nt=1 vs nt=0: median of the per-round ratios %.4f" % statistics.median([a / c for a, c in zip(res["1"], res["0"])]))
