"""what-if variants of the plane-fused 27-point relax pass on ONE solver (CEDAR_AMD_WHATIF is read per launch), interleaved
rounds: 0 = real kernel, 1 = no neighbour q rows, 2 = no k+1 slot-rows, 3 = both, 4 = no inter-plane slots, 5 = 4+1,
8 = k-pair walk (one launch: every run of rows in plane k, then the same rows of plane k-1; dependencies ignored).
CEDAR_AMD_FRUN sets the run length."""
import os, sys, json, statistics
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
os.environ["CEDAR_AMD_ILV"] = sys.argv[2] if len(sys.argv) > 2 else "0"
os.environ["CEDAR_AMD_NO_GRAPH"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cedar_amd import capi
so, b = capi.gallery("fe3", (n, n, n))
x = capi.DeviceArray(b.shape)
s = capi.Solver(so, share_operator=True)
WS = [0, 1, 2, 3, 4, 5, 8]
res = {w: [] for w in WS}
for rnd in range(5):
    for w in WS:
        os.environ["CEDAR_AMD_WHATIF"] = str(w)
        s.time_relax(x, b, 2)
        res[w].append(s.time_relax(x, b, 6) / 6)
for w in WS:
    print(json.dumps({"n": n, "ilv": os.environ["CEDAR_AMD_ILV"], "whatif": w, "median_ms_per_sweep": statistics.median(res[w]),
                      "min": min(res[w])}), flush=True)
