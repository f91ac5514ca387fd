"""sweep time of the plane-fused pass (CEDAR_AMD_WHATIF=0) against the k-pair what-if walk (8) at the run length
CEDAR_AMD_FRUN, interleaved on one solver"""
import os, sys, json, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CEDAR_AMD_ILV"] = "320"; os.environ["CEDAR_AMD_NO_GRAPH"] = "1"
from cedar_amd import capi
n = 512
so, b = capi.gallery("fe3", (n, n, n)); x = capi.DeviceArray(b.shape)
s = capi.Solver(so, share_operator=True)
res = {0: [], 8: []}
for rnd in range(4):
    for w in (0, 8):
        os.environ["CEDAR_AMD_WHATIF"] = str(w)
        s.time_relax(x, b, 2); res[w].append(s.time_relax(x, b, 6) / 6)
for w in (0, 8):
    print(json.dumps({"frun": os.environ.get("CEDAR_AMD_FRUN"), "whatif": w, "ms_per_sweep": statistics.median(res[w])}), flush=True)
