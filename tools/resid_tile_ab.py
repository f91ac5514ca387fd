"""interleaved A/B of the (j,k) tile shape the 27-point residual deals its rows in (CEDAR_AMD_TILE_RESID=tj,tk as log2,
read per call) on ONE solver: V-cycles without graph replay, many short rounds
    python tools/resid_tile_ab.py [n] [shapes ...]        # default 512, 4,4 8,8 2,8 8,2 6,6 1,1"""
import os, sys, json, statistics
os.environ["CEDAR_AMD_NO_GRAPH"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cedar_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
shapes = sys.argv[2:] or ["4,4", "8,8", "2,8", "8,2", "6,6", "1,1"]
so, b = capi.gallery("fe3", (n, n, n))
x = capi.DeviceArray(b.shape)
s = capi.Solver(so, share_operator=True)
res = {f: [] for f in shapes}
for rnd in range(8):
    for f in shapes:
        os.environ["CEDAR_AMD_TILE_RESID"] = f
        s.time_vcycles(x, b, 1)
        res[f].append(s.time_vcycles(x, b, 3) / 3)
for f in shapes:
    v = res[f]
    print(json.dumps({"n": n, "tile": f, "median_ms_per_vcycle": statistics.median(v), "min": min(v), "max": max(v)}), flush=True)
for f in shapes[1:]:
    print("%s vs %s: median of the per-round ratios %.4f" % (f, shapes[0], statistics.median([a / c for a, c in zip(res[f], res[shapes[0]])])))
